// Issue-rate probe for the integer VALU instructions of the RGB ingest kernels on gfx950: cycles per wave64 instruction
// (one SIMD) for v_perm_b32, v_dot2_u32_u16, v_mul_u32_u24, v_mul_hi_u32_u24, v_mad_u32_u24, v_mul_lo_u32, v_and_b32,
// v_add3_u32, v_lshrrev_b32.  16 independent chains per lane, 8 waves per SIMD, so the figure is the issue rate.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_int_rate tools/lab/valu_int_rate.hip && /tmp/valu_int_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(unsigned* out, int iters) {
    unsigned acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = out[threadIdx.x + i];
    const unsigned a = out[threadIdx.x + 17] | 1u, b = out[threadIdx.x + 18] | 3u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            if (MODE == 3) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            if (MODE == 4) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            if (MODE == 6) asm volatile("v_and_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            if (MODE == 7) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 8) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(acc[i]));
            if (MODE == 9) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(acc[i]) : "v"(a));
        }
    }
    unsigned t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

template <int MODE>
static void run(const char* name, unsigned* d) {
    const int iters = 4000, wgs = 256 * 8;                    // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(wgs), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(wgs), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)wgs * 4 * iters * 16 / 1024.0;
    printf("%-28s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD at 2.4 GHz (%.2f Ginstr/s per lane-chip)\n", name, ms,
           ms * 1e-3 * 2.4e9 / wave_instr_per_simd, (double)wgs * 256 * iters * 16 / ms * 1e-6);
}

int main() {
    unsigned* d;
    hipMalloc(&d, 256 * 8192 * sizeof(unsigned) + 256);
    hipMemset(d, 0x11, 256 * 8192 * sizeof(unsigned) + 256);
    run<0>("v_perm_b32", d);
    run<1>("v_dot2_u32_u16", d);
    run<2>("v_mul_u32_u24", d);
    run<3>("v_mul_hi_u32_u24", d);
    run<4>("v_mad_u32_u24", d);
    run<5>("v_mul_lo_u32", d);
    run<6>("v_and_b32", d);
    run<7>("v_add3_u32", d);
    run<8>("v_lshrrev_b32", d);
    run<9>("v_mul_u32_u24_sdwa", d);
    return 0;
}
