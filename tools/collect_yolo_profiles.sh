#!/bin/bash
# Refresh the YOLO-World backend's evidence only (subset of tools/collect_profiles.sh):  gpurun -- 'bash tools/collect_yolo_profiles.sh r02'
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PY="python $ROOT/bench.py"
rm -rf /tmp/prof_yolo
rocprofv3 --kernel-trace --stats -d /tmp/prof_yolo -o kt -- $PY --heuristic yolo --steps 48 --warmup 1 --no-cpu-baseline --no-verify --no-other-configs > "$OUT/${TAG}_bench_yolo_under_rocprofv3.json" 2> "$OUT/rocprof_yolo.err"
DBY=$(find /tmp/prof_yolo -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$DBY" --timed-region "$OUT/${TAG}_bench_yolo_under_rocprofv3.json" --check > "$OUT/${TAG}_yolo_rocprofv3_kernel_stats_timed_region.md" \
    || echo "collect_yolo_profiles: the kernel trace does NOT reproduce the bench line's conv launch average / count" >&2
rm -rf /tmp/prof_yolo2
rocprofv3 --kernel-trace -d /tmp/prof_yolo2 -o kt -- python $ROOT/tools/yolo_forward_probe.py 76 > "$OUT/${TAG}_yolo_forward_probe.log" 2>&1
python $ROOT/tools/rocpd_shapes.py "$(find /tmp/prof_yolo2 -name '*.db' | head -1)" > "$OUT/${TAG}_yolo_kernel_shapes_b76.md"
python $ROOT/tools/yolo_batch_sweep.py 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_yolo_batch_sweep.log"
# per-layer comparison of the conv kernels / halo forms at the chunk size, the grid-forward size and a small batch, and the
# per-layer shape / TFLOP/s tables under the launcher's own policy (the thresholds are read from these)
[ -z "${SKIP_HALO_FORMS:-}" ] && BATCHES="76 24 8" bash $ROOT/tools/yolo_halo_forms.sh $TAG
cd /tmp
rm -rf /tmp/prof_yolo3
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /tmp/prof_yolo3 -o pmc -- python $ROOT/tools/yolo_forward_probe.py 76 > /dev/null 2> "$OUT/rocprof_yolo_pmc.err"
python $ROOT/tools/rocpd_pmc.py "$(find /tmp/prof_yolo3 -name '*.db' | head -1)" > "$OUT/${TAG}_yolo_pmc_valu_by_kernel.md"
# fabric-side traffic of the convolutions, on the bench command itself (same launch population as its roofline.achieved)
# --steps 48: the default YOLO line's own launch population (two lock-step groups of 24); bench.py quotes the figure only for a run of the same population
YB="$PY --heuristic yolo --steps 48 --warmup 0 --no-cpu-baseline --no-verify --no-other-configs"
rm -rf /tmp/prof_yolo4
timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_yolo4 -o pmc -- $YB > "$OUT/pmc_pass_yolo_FETCH_SIZE.json" 2>> "$OUT/rocprof_yolo_pmc.err"
rm -rf /tmp/prof_yolo5
timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_yolo5 -o pmc -- $YB > /dev/null 2>> "$OUT/rocprof_yolo_pmc.err"
python $ROOT/tools/rocpd_traffic.py "$(find /tmp/prof_yolo4 -name '*.db' | head -1)" "$(find /tmp/prof_yolo5 -name '*.db' | head -1)" "conv_valu|conv_sw|conv_halo" \
    "rocprofv3 --kernel-trace --pmc <COUNTER> -- python bench.py --heuristic yolo --steps 48 --warmup 0 --no-cpu-baseline --no-verify --no-other-configs (tools/collect_yolo_profiles.sh)" "$OUT/pmc_pass_yolo_FETCH_SIZE.json" > "$OUT/${TAG}_yolo_pmc_conv_traffic.json"
cp "$OUT/${TAG}_yolo_pmc_conv_traffic.json" "$ROOT/profiles/${TAG}_yolo_pmc_conv_traffic.json"
$PY --heuristic yolo --steps 48 --warmup 1 > "$OUT/${TAG}_bench_yolo.json" 2>> "$OUT/bench_yolo.err"
ls -la "$OUT" | grep yolo
