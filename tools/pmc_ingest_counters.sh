#!/bin/bash
# SQ / TA counters of the RGB ingest kernels, generic form (TSTAR_INGEST_GENERIC=1, the round-2 kernels) against the fast path:
# VALU instructions per output pixel, wave-cycle breakdown, vector-memory instructions.   bash tools/pmc_ingest_counters.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for WHAT in grid resize; do
  for GEN in 1 0; do
    echo "== $WHAT, $([ $GEN = 1 ] && echo 'generic kernel (round 2)' || echo 'RGB fast path (round 3)')"
    i=0
    for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      rm -rf /tmp/pi$i
      TSTAR_INGEST_GENERIC=$GEN timeout 150 rocprofv3 --kernel-trace --pmc $C -d /tmp/pi$i -o pmc -- python $ROOT/tools/bench_ingest_one.py $WHAT 5 > /dev/null 2> /tmp/pi$i.err || tail -2 /tmp/pi$i.err
      python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pi$i -name '*.db' | head -1)" 2>/dev/null | grep -E "frames_to_grid|bilinear_gather" | head -3
    done
  done
done
