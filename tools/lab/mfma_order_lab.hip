// Lab: is a chain of v_mfma_f32_16x16x4_f32 bit-identical to the chain of v_mfma_f32_32x32x2_f32 the library's f32 tile runs, when the
// k indices are assigned to the k-slots so that every output element sees its products in the same order?
// Library order per 8 k (gemm_f32.hip): for s = 0..3: one 32x32x2 with slot 0 = k 8q+s, slot 1 = k 8q+4+s.
// 16x16x4 form under test: two instructions per 8 k with slots (8q+0, 8q+4, 8q+1, 8q+5) and (8q+2, 8q+6, 8q+3, 8q+7).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k32(const float* A, const float* W, float* C, int K, int N) {     // one wave: rows 0..31, cols 0..31
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int q = 0; q < K / 8; ++q)
        for (int s = 0; s < 4; ++s) {
            const int k = 8 * q + 4 * h + s;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k], W[l31 * K + k], acc, 0, 0, 0);
        }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * N + l31] = acc[r];
}
__global__ void k16(const float* A, const float* W, float* C, int K, int N) {     // one wave per 16x16 tile of the same 32x32 block
    const int lane = threadIdx.x, l15 = lane & 15, slot = lane >> 4;
    const int m0 = (blockIdx.x >> 1) * 16, n0 = (blockIdx.x & 1) * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < K / 8; ++q)
        for (int i = 0; i < 2; ++i) {
            // slots 0..3 of instruction i: k = 8q + 2i + {0, 4, 1, 5}
            const int k = 8 * q + 2 * i + (slot & 1) * 4 + (slot >> 1);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(m0 + l15) * K + k], W[(n0 + l15) * K + k], acc, 0, 0, 0);
        }
    for (int r = 0; r < 4; ++r) C[(m0 + 4 * slot + r) * N + n0 + l15] = acc[r];
}
int main() {
    const int K = 3072, N = 32;
    std::vector<float> a(32 * K), w(32 * K);
    unsigned s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f; };
    for (auto& v : a) v = rnd() * expf(3.f * rnd());
    for (auto& v : w) v = rnd() * 0.05f;
    float *dA, *dW, *dC1, *dC2;
    CK(hipMalloc(&dA, a.size() * 4)); CK(hipMalloc(&dW, w.size() * 4)); CK(hipMalloc(&dC1, 32 * N * 4)); CK(hipMalloc(&dC2, 32 * N * 4));
    CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dW, dC1, K, N);
    hipLaunchKernelGGL(k16, dim3(4), dim3(64), 0, 0, dA, dW, dC2, K, N);
    CK(hipDeviceSynchronize());
    std::vector<float> c1(32 * N), c2(32 * N);
    CK(hipMemcpy(c1.data(), dC1, c1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), dC2, c2.size() * 4, hipMemcpyDeviceToHost));
    int diff = 0; double maxd = 0;
    for (size_t i = 0; i < c1.size(); ++i) { if (c1[i] != c2[i]) ++diff; maxd = fmax(maxd, fabs((double)c1[i] - c2[i])); }
    // CPU fmaf chain in the library's k order
    int diff_cpu = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        float acc = 0.f;
        for (int q = 0; q < K / 8; ++q) for (int ss = 0; ss < 4; ++ss) { acc = fmaf(a[m * K + 8 * q + ss], w[n * K + 8 * q + ss], acc); acc = fmaf(a[m * K + 8 * q + 4 + ss], w[n * K + 8 * q + 4 + ss], acc); }
        if (acc != c1[m * N + n]) ++diff_cpu;
    }
    printf("16x16x4 chain vs 32x32x2 chain: %d of %zu elements differ (max |d| %.3e); 32x32x2 vs CPU fmaf chain in the same k order: %d differ\n", diff, c1.size(), maxd, diff_cpu);
    return 0;
}
