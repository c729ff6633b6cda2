# Counter passes over single variants of tools/lab/frag_lab (round 4): what the W-from-registers bf16-pipe loop waits on.
# Separate --pmc passes with --kernel-trace only.  Writes gpurun_out/frag_lab_counters.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/frag_lab_counters.md
cd /tmp
: > $OUT
for V in ${VARIANTS:-x9 x6 w2}; do
  for Z in 0 1; do
    echo "## $V N=3072 K=768 zero=$Z (un-profiled run first)" >> $OUT
    $ROOT/tools/lab/frag_lab one $V 3072 768 $Z 5 >> $OUT
    [ $Z = 1 ] && continue
    i=0
    for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
      i=$((i+1))
      rm -rf /tmp/pf$i
      timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pf$i -o pmc -- $ROOT/tools/lab/frag_lab one $V 3072 768 $Z 3 > /dev/null 2> /tmp/pf$i.err || tail -3 /tmp/pf$i.err
      DB=$(find /tmp/pf$i -name '*.db' | head -1)
      python $ROOT/tools/rocpd_pmc.py "$DB" | grep -E "gemm_frag" >> $OUT
    done
  done
done
cat $OUT
