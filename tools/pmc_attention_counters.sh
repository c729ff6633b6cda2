#!/bin/bash
# SQ wave-cycle breakdown of the attention kernels (separate --pmc passes): bash tools/pmc_attention_counters.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pa$i
  rocprofv3 --kernel-trace --pmc $C -d /tmp/pa$i -o pmc -- python $ROOT/tools/bench_attention.py > /dev/null 2> /tmp/pa$i.err || tail -3 /tmp/pa$i.err
  python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pa$i -name '*.db' | head -1)" | grep -E "attention_(f32|split)_kernel" | head -8
done
