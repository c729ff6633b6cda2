#!/bin/bash
# Collect the round's measured evidence on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh r01'
# Writes summaries under gpurun_out/profiles_<tag>/ (the raw rocpd databases stay on the box); copy the
# files to profiles/ afterwards.  Counter passes are separate runs with --kernel-trace only.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PY="python $ROOT/bench.py"

# 1. the secondary workloads (the bench line itself runs last, after the counter passes, so that its
#    roofline.traffic field is this collection's figure)
: > "$OUT/bench.err"
$PY --steps 32 --warmup 1 --grid 4 --lockstep 16 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_grid4_lockstep16.json" 2>> "$OUT/bench.err"
$PY --steps 8 --warmup 1 --weights bf16 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_bf16_weights.json" 2>> "$OUT/bench.err"
$PY --steps 8 --warmup 1 --weights f32 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_f32_native.json" 2>> "$OUT/bench.err"
$PY --steps 8 --warmup 1 --weights bf16 --nframes 14400 --grid 15 --search-nframes 32 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_config5.json" 2>> "$OUT/bench.err"
$PY --steps 8 --warmup 1 --weights bf16_exact --nframes 14400 --grid 15 --search-nframes 32 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_config5_exact_split.json" 2>> "$OUT/bench.err"
$PY --steps 8 --warmup 1 --concurrency 2 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_concurrency2.json" 2>> "$OUT/bench.err"
$PY --steps 8 --warmup 1 --workload haystack --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_haystack_1gpu.json" 2>> "$OUT/bench.err"
$PY --workload haystack32 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_haystack32_1gpu.json" 2>> "$OUT/bench.err"

# 1b. YOLO-World backend: its own script (bench line, kernel trace, per-shape and per-layer tables, counters)
bash $ROOT/tools/collect_yolo_profiles.sh $TAG
cd /tmp

# 2. kernel trace of the bench command (the driver's default flags: the untimed post-timer legs -- keyframe
#    verification, config.grid4 -- run too; the timed region is cut out of the trace at the marker kernels bench.py
#    enqueues around it, tools/rocpd_window.py)
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $PY --steps 16 --warmup 1 --no-cpu-baseline --no-other-configs --no-drop-in > "$OUT/${TAG}_bench_under_rocprofv3.json" 2> "$OUT/rocprof_kt.err"
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
BJ="$OUT/${TAG}_bench_under_rocprofv3.json"
python $ROOT/tools/rocpd_stats.py "$DB" > "$OUT/${TAG}_rocprofv3_kernel_stats.md"
# --check: the trace must reproduce the bench line's roofline leg (average GEMM launch within 3 %, dispatch count within 2 %)
if ! python $ROOT/tools/rocpd_stats.py "$DB" --timed-region "$BJ" --check > "$OUT/${TAG}_rocprofv3_kernel_stats_timed_region.md"; then
    echo "collect_profiles: the kernel trace does NOT reproduce roofline.avg_launch_ms / launches_timed -- see ${TAG}_rocprofv3_kernel_stats_timed_region.md" >&2
    FAILED=1
fi
python $ROOT/tools/rocpd_shapes.py "$DB" > "$OUT/${TAG}_rocprofv3_kernel_shapes.md"
python $ROOT/tools/rocpd_gaps.py "$DB" --timed-region "$BJ" > "$OUT/${TAG}_gpu_idle_gaps.txt"

# 3. counter passes (one counter set per run)
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES"; do
    N=$(echo $C | tr ' ' '_')
    rm -rf /tmp/prof_$N
    # --steps 16: the default line's own launch population (two lock-step groups of 8), so that bytes per launch compare like with like;
    # --no-grid4 --no-verify: every GEMM dispatch of the run is one of the timed steps' (plus the text tower's 48 tiny ones),
    # so the per-launch averages are over the same population as the bench line's avg_launch_gflop
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof_$N -o pmc -- $PY --steps 16 --warmup 0 --no-cpu-baseline --no-grid4 --no-verify --no-other-configs --no-drop-in > "$OUT/pmc_pass_$N.json" 2> "$OUT/rocprof_$N.err"
done
F=$(find /tmp/prof_FETCH_SIZE -name '*.db' | head -1)
W=$(find /tmp/prof_WRITE_SIZE -name '*.db' | head -1)
M=$(find /tmp/prof_SQ_VALU_MFMA_BUSY_CYCLES_GRBM_GUI_ACTIVE -name '*.db' | head -1)
M2=$(find /tmp/prof_SQ_INSTS_VALU_MFMA_MOPS_F32_SQ_BUSY_CYCLES -name '*.db' | head -1)
python $ROOT/tools/rocpd_traffic.py "$F" "$W" "gemm_f32|gemm_bf16w2_wide" "" "$OUT/pmc_pass_FETCH_SIZE.json" > "$OUT/${TAG}_pmc_gemm_traffic_f32x3.json"      # the bench default is the f32x3 mode since round 5
python $ROOT/tools/rocpd_pmc.py "$F" "$W" > "$OUT/${TAG}_pmc_fetch_write_by_kernel.md"
# the same two counters on the DRIVER's launch population (--steps 20: two lock-step groups of 10), so that the driver's line can quote a like-for-like figure too
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof20_$C
    timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/prof20_$C -o pmc -- $PY --steps 20 --warmup 0 --no-cpu-baseline --no-grid4 --no-verify --no-other-configs --no-drop-in > "$OUT/pmc_pass20_$C.json" 2> "$OUT/rocprof20_$C.err"
done
python $ROOT/tools/rocpd_traffic.py "$(find /tmp/prof20_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/prof20_WRITE_SIZE -name '*.db' | head -1)" "gemm_f32|gemm_bf16w2_wide" \
    "rocprofv3 --kernel-trace --pmc <COUNTER> -- python bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-grid4 --no-verify --no-other-configs --no-drop-in (tools/collect_profiles.sh)" \
    "$OUT/pmc_pass20_FETCH_SIZE.json" > "$OUT/${TAG}_pmc_gemm_traffic_f32x3_steps20.json"
cp "$OUT/${TAG}_pmc_gemm_traffic_f32x3_steps20.json" "$ROOT/profiles/${TAG}_pmc_gemm_traffic_f32x3_steps20.json"
python $ROOT/tools/rocpd_pmc.py "$M" "$M2" > "$OUT/${TAG}_pmc_mfma_by_kernel.md"
python $ROOT/tools/rocpd_mfma.py "$M" > "$OUT/${TAG}_pmc_mfma_utilisation.md"

# 4. the bench line (default flags), reading the traffic figure just collected
cp "$OUT/${TAG}_pmc_gemm_traffic_f32x3.json" "$ROOT/profiles/${TAG}_pmc_gemm_traffic_f32x3.json"
$PY > "$OUT/${TAG}_bench.json" 2>> "$OUT/bench.err"
$PY --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_shape.json" 2>> "$OUT/bench.err"

# 4b. one reference-default 4x4 search alone (the drop-in user's latency): A/Bs, kernel table, idle gaps, per-stream view
bash $ROOT/tools/solo_grid4_trace.sh > "$OUT/solo_trace.log" 2>&1
for f in solo_grid4_probe.log solo_grid4_kernel_stats.md solo_grid4_gaps.txt solo_grid4_streams.txt; do
    [ -f "$ROOT/gpurun_out/$f" ] && cp "$ROOT/gpurun_out/$f" "$OUT/${TAG}_$f"
done
cd /tmp

# 5. kernel microbenchmarks
rm -rf /tmp/prof_sf
rocprofv3 --kernel-trace -d /tmp/prof_sf -o sf -- python $ROOT/tools/small_forward_probe.py 40 > "$OUT/${TAG}_small_forward_probe.log" 2>&1
python $ROOT/tools/rocpd_gaps.py "$(find /tmp/prof_sf -name '*.db' | head -1)" 0.9 > "$OUT/${TAG}_small_forward_idle_gaps.txt"
python $ROOT/tools/small_forward_probe.py 40 f32x3 2>&1 | grep -v amdgpu.ids >> "$OUT/${TAG}_small_forward_probe.log"      # the bench's mode (the traced run above is the library default, f32)
[ -x $ROOT/tools/lab/pkfma_rate ] && $ROOT/tools/lab/pkfma_rate > "$OUT/${TAG}_valu_pkfma_rate.log" 2>&1
python $ROOT/tools/bench_gemm_cfg.py > "$OUT/${TAG}_gemm_tile_configs.log" 2>&1
python $ROOT/tools/sweep_small_m.py 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_gemm_subwave_sweep.log"
python $ROOT/tools/bench_kernels.py > "$OUT/${TAG}_kernel_microbench.log" 2>&1
python $ROOT/tools/bench_ingest.py 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_ingest_bandwidth.log"
python $ROOT/tools/bench_gemm_bf16.py 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_gemm_bf16_tiles.log"
python $ROOT/tools/bench_attention.py 16 64 256 2>&1 | grep -v amdgpu.ids > "$OUT/${TAG}_attention_microbench.log"
[ -x $ROOT/tools/lab/valu_int_rate ] && $ROOT/tools/lab/valu_int_rate > "$OUT/${TAG}_valu_int_issue_rate.log" 2>&1
ls -la "$OUT"
exit ${FAILED:-0}
