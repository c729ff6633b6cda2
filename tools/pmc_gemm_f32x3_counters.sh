# The f32x3 GEMM tile INSIDE the library under the bench (round 4): matrix-pipe busy cycles, core clock (GRBM_GUI_ACTIVE / duration),
# wave-cycle breakdown, LDS activity -- next to the native f32 tile in the same run shape.  Separate --pmc passes with --kernel-trace only.
# Writes gpurun_out/gemm_f32x3_counters.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/gemm_f32x3_counters.md
cd /tmp
: > $OUT
for W in f32x3 f32; do
  echo "## --weights $W" >> $OUT
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rm -rf /tmp/px$i
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/px$i -o pmc -- python $ROOT/bench.py --steps 2 --warmup 0 --lockstep 2 --pipeline 1 --weights $W --no-cpu-baseline --no-grid4 --no-verify --no-other-configs > /dev/null 2> /tmp/px$i.err || tail -3 /tmp/px$i.err
    DB=$(find /tmp/px$i -name '*.db' | head -1)
    python $ROOT/tools/rocpd_pmc.py "$DB" | grep -E "gemm_bf16w2_wide_kernel<4, 1, true, false|gemm_bf16w2_wide_kernel<4, 0, true, false, false|hybrid_kernel<0, 1, true, false|hybrid_kernel<0, 0, true, false, false" >> $OUT
  done
done
cat $OUT
