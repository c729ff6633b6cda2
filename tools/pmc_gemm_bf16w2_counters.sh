# What the two-term bf16-weight GEMM tile (gemm_bf16w2_wide_kernel, configs[4]) waits on, next to the exact-f32 tile:
# matrix-pipe busy cycles, LDS instruction activity / bank conflicts / waits, wave wait share.  Separate --pmc passes with
# --kernel-trace only.  Writes gpurun_out/gemm_bf16w2_counters.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/gemm_bf16w2_counters.md
cd /tmp
: > $OUT
for W in bf16 f32; do
  echo "## --weights $W" >> $OUT
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"; do
    i=$((i+1))
    rm -rf /tmp/pw$i
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pw$i -o pmc -- python $ROOT/bench.py --steps 2 --warmup 0 --lockstep 2 --pipeline 1 --weights $W --no-cpu-baseline --no-grid4 --no-verify > /dev/null 2> /tmp/pw$i.err || tail -3 /tmp/pw$i.err
    DB=$(find /tmp/pw$i -name '*.db' | head -1)
    python $ROOT/tools/rocpd_pmc.py "$DB" | grep -E "gemm_bf16w2_wide_kernel<0, true, false|gemm_bf16w2_wide_kernel<1, true, false|hybrid_kernel<0, 0, true, false|hybrid_kernel<0, 1, true, false" >> $OUT
  done
done
cat $OUT
