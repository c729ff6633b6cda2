set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for T16 in 0 1; do
  echo "== TSTAR_ATTN_T16=$T16"
  i=0
  for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    i=$((i+1)); rm -rf /tmp/pa$i
    TSTAR_ATTN_T16=$T16 timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pa$i -o pmc -- python $ROOT/tools/bench_attention.py 256 > /dev/null 2> /tmp/pa$i.err || tail -3 /tmp/pa$i.err
    python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pa$i -name '*.db' | head -1)" | grep -E "attention_f32_kernel" | head -8
  done
done
