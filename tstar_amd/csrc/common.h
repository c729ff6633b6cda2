// Shared helpers for the tstar_hip C-ABI library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace tstar {

// last error string, thread-local; returned by tstar_last_error()
void set_error(const std::string& msg);

#define TSTAR_HIP_CHECK(expr)                                                        \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) {                                                      \
            ::tstar::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));   \
            return TSTAR_ERR_HIP;                                                    \
        }                                                                            \
    } while (0)

#define TSTAR_REQUIRE(cond, msg)                                                     \
    do {                                                                             \
        if (!(cond)) {                                                               \
            ::tstar::set_error(std::string(msg) + " (" #cond ")");                  \
            return TSTAR_ERR_ARG;                                                    \
        }                                                                            \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that need more than 64 KB of dynamic LDS: set once per
// (kernel, device) under a mutex -- launches may come from several host threads and a process may use more than
// one device (the attribute is applied to the kernel's code object of the CURRENT device).  Returns TSTAR_OK / TSTAR_ERR_HIP.
int ensure_dyn_lds(const void* kernel, int bytes);

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Bijective XCD-aware remap of a 1-D block id: the dispatcher places block b on
// XCD b % 8 (observed, speed only); give each XCD a contiguous chunk of the
// tile space so neighbouring tiles share that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace tstar

#define TSTAR_OK 0
#define TSTAR_ERR_ARG 1
#define TSTAR_ERR_HIP 2
#define TSTAR_ERR_STATE 3
