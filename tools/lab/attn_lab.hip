// Lab (round 5): the vision tower's attention on the bf16 matrix pipe with EXACT three-term operands (the f32x3 scheme of the
// GEMM: every f32 operand as three round-to-nearest bf16 terms, the six partial products with i + j <= 2, f32 accumulation),
// software-pipelined per wave.  `attention_x3.h` holds the kernel (the library includes the same file); this driver checks it
// against float64 next to a plain f32 VALU reference and times it at the bench shape (B images x 577 tokens x 12 heads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <utility>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define TSTAR_ATTN_X3_LAB 1          // also compile the plane-tile + LDS-DMA form (not adopted by the library)
#include "../../tstar_amd/csrc/attention_x3.h"

static void attn_ref(const std::vector<float>& qkv, std::vector<double>& out, int B, int T, int heads) {
    const int D = heads * 64, D3 = 3 * D;
    out.assign((size_t)B * T * D, 0.0);
    std::vector<double> sc(T);
    for (int b = 0; b < B; ++b)
        for (int hd = 0; hd < heads; ++hd)
            for (int q = 0; q < T; ++q) {
                const float* qp = &qkv[((size_t)b * T + q) * D3 + hd * 64];
                double mx = -1e300;
                for (int k = 0; k < T; ++k) {
                    const float* kp = &qkv[((size_t)b * T + k) * D3 + D + hd * 64];
                    double s = 0;
                    for (int d = 0; d < 64; ++d) s += (double)qp[d] * (double)kp[d];
                    sc[k] = s * 0.125;
                    mx = fmax(mx, sc[k]);
                }
                double l = 0;
                for (int k = 0; k < T; ++k) { sc[k] = exp(sc[k] - mx); l += sc[k]; }
                double* op = &out[((size_t)b * T + q) * D + hd * 64];
                for (int k = 0; k < T; ++k) {
                    const float* vp = &qkv[((size_t)b * T + k) * D3 + 2 * D + hd * 64];
                    const double p = sc[k] / l;
                    for (int d = 0; d < 64; ++d) op[d] += p * (double)vp[d];
                }
            }
}

int main(int argc, char** argv) {
    unsigned s = 777;
    auto uni = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFFFF) / 16777216.0f; };
    auto gauss = [&]() { float u1 = uni() + 1e-9f, u2 = uni(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    // ---- correctness on small problems (T = 577 and ragged lengths), peaky softmax
    struct Case { int B, T, heads; } cases[] = {{1, 577, 2}, {2, 97, 3}, {1, 33, 2}, {1, 128, 1}, {1, 1, 1}, {1, 600, 2}, {1, 64, 1}, {1, 65, 1}};
    for (auto c : cases) {
        const int D = c.heads * 64;
        std::vector<float> qkv((size_t)c.B * c.T * 3 * D);
        for (auto& v : qkv) v = gauss();
        for (int r = 0; r < c.B * c.T; ++r) for (int d = 0; d < D; ++d) qkv[(size_t)r * 3 * D + d] *= 3.0f;
        if (c.T > 500) for (int d = 0; d < D; ++d) { qkv[(size_t)500 * 3 * D + D + d] *= 10.f; qkv[(size_t)(c.T - 1) * 3 * D + D + d] *= 12.f; }
        std::vector<double> ref;
        attn_ref(qkv, ref, c.B, c.T, c.heads);
        float *dq, *dout;
        CK(hipMalloc(&dq, qkv.size() * 4)); CK(hipMalloc(&dout, ref.size() * 4));
        CK(hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dout, 0xFF, ref.size() * 4));
        char* dpl; CK(hipMalloc(&dpl, tstar::kv_planes_bytes(c.B * c.T, c.heads)));
        CK(hipMemset(dpl, 0xFF, tstar::kv_planes_bytes(c.B * c.T, c.heads)));
        for (int variant = 0; variant < 2; ++variant) {
            CK(hipMemset(dout, 0xFF, ref.size() * 4));
            if (variant == 1 && tstar::kv_planes_launch(dq, dpl, c.B * c.T, c.heads, 0) != 0) { printf("planes launch failed\n"); return 1; }
            if (tstar::attention_x3_launch(dq, dout, c.B, c.T, c.heads, 0, variant ? dpl : nullptr) != 0) { printf("launch failed\n"); return 1; }
            CK(hipDeviceSynchronize());
            std::vector<float> o(ref.size());
            CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, se = 0; bool nan = false;
            for (size_t i = 0; i < o.size(); ++i) { const double e = o[i] - ref[i]; if (!(e == e)) nan = true; maxe = fmax(maxe, fabs(e)); se += e * e; }
            printf("B=%d T=%3d heads=%d %s: max|err| %.3e  rms %.3e%s\n", c.B, c.T, c.heads, variant ? "plane tiles + LDS-DMA" : "in-kernel staging    ", maxe,
                   sqrt(se / o.size()), (nan || maxe > 5e-5) ? "   <-- WRONG" : "");
        }
        CK(hipFree(dq)); CK(hipFree(dout)); CK(hipFree(dpl));
    }
    // ---- speed at the bench shape
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = 577, heads = 12, D = heads * 64;
    std::vector<float> qkv((size_t)B * T * 3 * D);
    for (auto& v : qkv) v = gauss();
    float *dq, *dout;
    CK(hipMalloc(&dq, qkv.size() * 4)); CK(hipMalloc(&dout, (size_t)B * T * D * 4));
    CK(hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double fl = 4.0 * B * heads * (double)T * T * 64;
    char* dpl; CK(hipMalloc(&dpl, tstar::kv_planes_bytes(B * T, heads)));
    tstar::kv_planes_launch(dq, dpl, B * T, heads, 0);
    for (int rep = 0; rep < 6; ++rep) {
        const char* pl = (rep & 1) ? dpl : nullptr;
        for (int i = 0; i < 2; ++i) tstar::attention_x3_launch(dq, dout, B, T, heads, 0, pl);
        CK(hipEventRecord(e0));
        const int it = 5;
        for (int i = 0; i < it; ++i) tstar::attention_x3_launch(dq, dout, B, T, heads, 0, pl);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
        printf("attention_x3 %s B=%d T=%d heads=%d: %.3f ms  %.1f TFLOP/s algorithmic  %.0f executed (x6)\n", pl ? "plane tiles + LDS-DMA" : "in-kernel staging    ",
               B, T, heads, ms, fl / ms / 1e9, 6 * fl / ms / 1e9);
    }
    {   // the stand-alone plane pass, for scale (inside the model the qkv GEMM's epilogue writes the tiles)
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) tstar::kv_planes_launch(dq, dpl, B * T, heads, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kv_planes_kernel (stand-alone split pass): %.3f ms\n", ms / 5);
    }
    return 0;
}
