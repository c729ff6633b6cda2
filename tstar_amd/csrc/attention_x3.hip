// Library entry of the f32x3 mode's vision attention; the kernel lives in attention_x3.h (shared with tools/lab/attn_lab.hip).
#include "common.h"
#include "kernels.h"
#include "prof.h"
#include "attention_x3.h"

namespace tstar {

int attention_x3(const float* qkv, float* out, int B, int T, int heads, hipStream_t s) {
    TSTAR_REQUIRE(B > 0 && T > 0 && heads > 0, "attention_x3: empty problem");
    TSTAR_REQUIRE(((size_t)T + 32) * 3 * heads * 64 * 4 < (1ull << 31), "attention_x3: one image's qkv rows (plus one key tile of look-ahead) must stay below 2 GiB");
    // 69 KB of dynamic LDS: the limit is a per-device attribute of the kernel (a second OWLInterface on cuda:1 needs it set there too)
    if (int rc = ensure_dyn_lds(reinterpret_cast<const void*>(ax3::attention_x3_kernel<false>), ax3::LDS_BYTES)) return rc;
    const bool prof = prof_enabled();
    if (prof) prof_start(PROF_ATTN, s, 4.0 * B * heads * (double)T * T * 64);
    const int rc = attention_x3_launch(qkv, out, B, T, heads, s);
    if (prof) prof_stop(PROF_ATTN, s);
    TSTAR_HIP_CHECK((hipError_t)rc);
    return TSTAR_OK;
}

}  // namespace tstar
