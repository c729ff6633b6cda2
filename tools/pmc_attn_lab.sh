# Counter passes over tools/lab/attn_lab (round 5): what the pipelined f32x3 attention waits on.  Separate --pmc passes, --kernel-trace only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/r05_attention_x3_counters.md
cd /tmp
echo "# attention_x3_kernel (tools/lab/attn_lab 256: B = 256, T = 577, 12 heads), rocprofv3 --pmc passes; <false> = in-kernel staging (the library's), <true> = plane tiles + LDS-DMA" > $OUT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pa$i
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pa$i -o pmc -- $ROOT/tools/lab/attn_lab 256 > /dev/null 2> /tmp/pa$i.err || tail -3 /tmp/pa$i.err
  DB=$(find /tmp/pa$i -name '*.db' | head -1)
  python $ROOT/tools/rocpd_pmc.py "$DB" | grep -E "attention_x3_kernel" >> $OUT
done
cat $OUT
