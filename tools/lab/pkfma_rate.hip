// Issue-rate probe for v_pk_fma_f32 on gfx950: what the f32 VALU sustains with (a) three VGPR-pair operands, (b) an SGPR
// pair as the multiplier, (c) the op_sel broadcast of one half of src0 -- the forms the YOLO conv kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pkfma_rate tools/lab/pkfma_rate.hip && /tmp/pkfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float s0, float s1) {
    f32x2 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x2{(float)threadIdx.x, (float)i};
    f32x2 a = {out[threadIdx.x], out[threadIdx.x + 1]};
    f32x2 wv = {out[threadIdx.x + 2], out[threadIdx.x + 3]};
    const f32x2 ws = {s0, s1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(wv));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(ws));
            if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "s"(ws));
            if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(a), "s"(ws));
            if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(wv.x));
        }
    }
    f32x2 t = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) t += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}

template <int MODE>
static void run(const char* name, float* d, int wgs) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(wgs), dim3(256), 0, 0, d, 10, 1.0f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * 256 * iters * 32 * (MODE == 4 ? 2 : 4);
    printf("%-44s wgs %5d: %8.3f ms  %7.1f TFLOP/s\n", name, wgs, ms, flops / ms * 1e-9);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 8192 * sizeof(float) + 64);
    hipMemset(d, 0, 256 * 8192 * sizeof(float) + 64);
    for (int wgs : {256 * 2, 256 * 4, 256 * 8}) {
        run<0>("v_pk_fma_f32 v, v, v", d, wgs);
        run<1>("v_pk_fma_f32 v, v, s", d, wgs);
        run<2>("v_pk_fma_f32 v, v.lo (op_sel), s", d, wgs);
        run<3>("v_pk_fma_f32 v, v.hi (op_sel), s", d, wgs);
        run<4>("v_fma_f32 v, v, v", d, wgs);
    }
    return 0;
}
