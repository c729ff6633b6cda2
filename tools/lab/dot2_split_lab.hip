// Lab (round 6): the exact three-term bf16 split  a = a0 + a1 + a2  (a0 = bf16_rn(a), a1 = bf16_rn(a - a0), a2 = bf16_rn(a - a0 - a1))
// with the remainders formed by v_dot2c_f32_bf16 -- r = dot2((a0.lo, a0.hi), (-1, 0)) + x -- instead of a shift / mask + v_sub_f32:
// 3 cvt_pk + 4 dot2 = 7 VALU per PAIR of floats against 3 cvt_pk + 4 shift/and + 4 sub = 11.  Questions: (1) are the terms bit-identical
// to the subtract form (the subtraction is exact in f32, so any correctly rounded fused form must agree -- unless the dot unit flushes
// or truncates), over all binades; (2) what does a wave issue per cycle of either form.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 dot2_split_lab.hip -o dot2_split_lab && ./dot2_split_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void split_sub(float x0, float x1, unsigned (&o)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f32x2 x; x[0] = x0; x[1] = x1;
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
        o[k] = hb;
        if (k < 2) { x0 = x0 - __uint_as_float(hb << 16); x1 = x1 - __uint_as_float(hb & 0xFFFF0000u); }
    }
}
__device__ __forceinline__ void split_dot2(float x0, float x1, unsigned (&o)[3]) {
    const bf16x2 nlo = __builtin_bit_cast(bf16x2, 0x0000BF80u), nhi = __builtin_bit_cast(bf16x2, 0xBF800000u);   // (-1, 0), (0, -1)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f32x2 x; x[0] = x0; x[1] = x1;
        const bf16x2 b = __builtin_convertvector(x, bf16x2);
        o[k] = __builtin_bit_cast(unsigned, b);
        if (k < 2) { x0 = __builtin_amdgcn_fdot2_f32_bf16(b, nlo, x0, false); x1 = __builtin_amdgcn_fdot2_f32_bf16(b, nhi, x1, false); }
    }
}
__global__ void both_kernel(const float* x, unsigned* a, unsigned* b, size_t npairs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    unsigned o[3];
    split_sub(x[2 * i], x[2 * i + 1], o);
    a[3 * i] = o[0]; a[3 * i + 1] = o[1]; a[3 * i + 2] = o[2];
    split_dot2(x[2 * i], x[2 * i + 1], o);
    b[3 * i] = o[0]; b[3 * i + 1] = o[1]; b[3 * i + 2] = o[2];
}
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + 17 * i);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            unsigned o[3];
            if (MODE == 0) split_sub(v[2 * p], v[2 * p + 1], o); else split_dot2(v[2 * p], v[2 * p + 1], o);
            acc ^= o[0] + o[1] * 3u + o[2] * 5u;
            v[2 * p] += 1.0f; v[2 * p + 1] -= 0.5f;       // new operands every round (2 VALU per pair in both modes)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(acc);
}

int main() {
    const size_t npairs = (size_t)1 << 24;
    std::vector<float> h(2 * npairs);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (size_t i = 0; i < 2 * npairs; ++i) {
        unsigned bits = rnd() & 0x807FFFFFu;
        unsigned e;
        const unsigned cls = (unsigned)(i % 8);
        if (cls == 0) e = rnd() % 255;                 // any finite binade incl. denormals (e = 0) 
        else if (cls == 1) e = 1 + rnd() % 24;         // terms go denormal
        else if (cls == 2) e = 230 + rnd() % 25;       // huge
        else e = 97 + rnd() % 60;                      // the activations' range
        if (cls == 3) bits = (bits & 0x807F0000u) | 0x8000u;       // ties of the first rounding
        bits |= e << 23;
        memcpy(&h[i], &bits, 4);
    }
    float* dx; unsigned *da, *db;
    CK(hipMalloc(&dx, h.size() * 4)); CK(hipMalloc(&da, npairs * 12)); CK(hipMalloc(&db, npairs * 12));
    CK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    both_kernel<<<(unsigned)((npairs + 255) / 256), 256>>>(dx, da, db, npairs);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> a(3 * npairs), b(3 * npairs);
    CK(hipMemcpy(a.data(), da, npairs * 12, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, npairs * 12, hipMemcpyDeviceToHost));
    size_t diff_all = 0, diff_normal = 0, n_normal = 0, shown = 0;
    for (size_t i = 0; i < npairs; ++i) {
        // "terms stay normal": |x| >= 2^-100 for both members of the pair (the third term is >= 2^-24 |x| when non-zero)
        unsigned b0, b1; memcpy(&b0, &h[2 * i], 4); memcpy(&b1, &h[2 * i + 1], 4);
        const bool normal = ((b0 >> 23) & 255) >= 27 && ((b1 >> 23) & 255) >= 27;
        n_normal += normal;
        const bool d = a[3 * i] != b[3 * i] || a[3 * i + 1] != b[3 * i + 1] || a[3 * i + 2] != b[3 * i + 2];
        diff_all += d; diff_normal += d && normal;
        if (d && shown < 6) { printf("  diff: x = (%08x, %08x)  sub %08x %08x %08x  dot2 %08x %08x %08x\n", b0, b1, a[3*i], a[3*i+1], a[3*i+2], b[3*i], b[3*i+1], b[3*i+2]); ++shown; }
    }
    printf("pairs %zu: %zu differ in all; %zu differ among the %zu pairs whose terms stay normal (|x| >= 2^-100)\n", npairs, diff_all, diff_normal, n_normal);
    // ---- issue rate
    float* dout; CK(hipMalloc(&dout, 256 * 2048 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) rate_kernel<0><<<2048, 256>>>(dout, iters); else rate_kernel<1><<<2048, 256>>>(dout, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%s: %.3f ms for %.3g pair-splits -> %.1f G pair-splits/s\n", mode ? "dot2 form (7 VALU/pair + 5 bookkeeping)" : "sub form (11 VALU/pair + 5 bookkeeping)",
                            ms, 2048.0 * 256 * iters * 8, 2048.0 * 256 * iters * 8 / ms / 1e6);
        }
    }
    return 0;
}
