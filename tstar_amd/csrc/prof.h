// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
#pragma once
#include "common.h"

namespace tstar {
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_CONV = 2, PROF_NCAT = 3 };   // PROF_CONV: conv_valu_kernel (YOLO-World backend)
bool prof_enabled();
// record the start / stop events around one launch; `work` = algorithmic flops of the launch, `bytes` = its algorithmic
// HBM bytes (every operand read once, the result written once)
void prof_start(int cat, hipStream_t s, double work, double bytes = 0.0);
void prof_stop(int cat, hipStream_t s);
}  // namespace tstar
