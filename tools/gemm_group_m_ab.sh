# Tile order A/B (round 4): TSTAR_GEMM_GM = rows of a super-panel in gemm_f32.hip's tile_mn (1 = panel-major, rounds 1-3).
# Per value: the un-profiled bench line (GEMM TFLOP/s, frames/s) and the fabric-side traffic per GEMM launch from separate
# FETCH_SIZE / WRITE_SIZE passes (same command as tools/collect_profiles.sh).  Writes gpurun_out/gemm_group_m_ab.md
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
OUT=$ROOT/gpurun_out/gemm_group_m_ab.md
cd /tmp
echo "| TSTAR_GEMM_GM | weights | frames/s | GEMM TFLOP/s (events) | fabric bytes per launch (FETCH x2 + WRITE) | algorithmic bytes per launch | ratio |" > $OUT
echo "|---:|---|---:|---:|---:|---:|---:|" >> $OUT
for W in ${WEIGHTS:-f32 f32x3}; do
for GM in ${GMS:-1 4 8 16}; do
  export TSTAR_GEMM_GM=$GM
  python $ROOT/bench.py --weights $W --steps 8 --warmup 1 --no-cpu-baseline --no-grid4 --no-other-configs > /tmp/gm_b.json 2> /tmp/gm_b.err || tail -3 /tmp/gm_b.err
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/gm_$C
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/gm_$C -o pmc -- python $ROOT/bench.py --weights $W --steps 4 --warmup 0 --no-cpu-baseline --no-grid4 --no-verify --no-other-configs > /dev/null 2> /tmp/gm_$C.err || tail -3 /tmp/gm_$C.err
  done
  F=$(find /tmp/gm_FETCH_SIZE -name '*.db' | head -1)
  Wd=$(find /tmp/gm_WRITE_SIZE -name '*.db' | head -1)
  python $ROOT/tools/rocpd_traffic.py "$F" "$Wd" gemm > /tmp/gm_t.json
  python - >> $OUT <<PY
import json
b=json.load(open('/tmp/gm_b.json')); t=json.load(open('/tmp/gm_t.json')); r=b['roofline']
print(f"| $GM | $W | {b['value']:.0f} | {r['achieved_algorithmic']:.1f} | {t['bytes_per_launch_corrected']:.4g} | {r['algorithmic_bytes_per_launch']:.4g} | {t['bytes_per_launch_corrected']/r['algorithmic_bytes_per_launch']:.2f} |")
PY
done
done
cat $OUT
