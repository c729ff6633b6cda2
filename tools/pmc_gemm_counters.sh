set -u
ROOT=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/p$i
  rocprofv3 --kernel-trace --pmc $C -d /tmp/p$i -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --lockstep 1 --weights ${1:-f32x3} --no-cpu-baseline > /dev/null 2> /tmp/p$i.err || tail -3 /tmp/p$i.err
  DB=$(find /tmp/p$i -name '*.db' | head -1)
  python $ROOT/tools/rocpd_pmc.py "$DB" | grep -E "gemm_f32_kernel<Cfg<2, 2, 2>, [0-9], 1, true|hybrid_kernel<[0-9], 0, true, true" | head -12
done
