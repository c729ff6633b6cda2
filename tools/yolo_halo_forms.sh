#!/bin/bash
# Per-layer comparison of the halo forms (16 / 8 channels per wave, both patch shapes) against the other conv kernels at a few
# batch sizes: the table the launcher's HALO8_MAX / HALO8_MIN thresholds are read from.   gpurun -- 'bash tools/yolo_halo_forms.sh r03'
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for B in ${BATCHES:-76 38 16 8}; do
  for V in "nohalo 0 0" "h16 2 16" "h8 2 8" "policy -1 0"; do
    set -- $V
    rm -rf /tmp/hf_$1
    if [ "$1" = policy ]; then
      rocprofv3 --kernel-trace -d /tmp/hf_$1 -o kt -- python $ROOT/tools/yolo_forward_probe.py $B > /dev/null 2>&1
    else
      TSTAR_YOLO_HALO=$2 TSTAR_YOLO_HALO_NCH=$3 rocprofv3 --kernel-trace -d /tmp/hf_$1 -o kt -- python $ROOT/tools/yolo_forward_probe.py $B > /dev/null 2>&1
    fi
  done
  python $ROOT/tools/rocpd_conv_align.py 7 nohalo=$(find /tmp/hf_nohalo -name '*.db' | head -1) h16=$(find /tmp/hf_h16 -name '*.db' | head -1) \
      h8=$(find /tmp/hf_h8 -name '*.db' | head -1) policy=$(find /tmp/hf_policy -name '*.db' | head -1) > "$OUT/${TAG}_yolo_halo_forms_by_layer_b$B.md"
  python $ROOT/tools/yolo_layer_table.py $(find /tmp/hf_policy -name '*.db' | head -1) $B 7 > "$OUT/${TAG}_yolo_layers_b$B.md"
  tail -1 "$OUT/${TAG}_yolo_halo_forms_by_layer_b$B.md"
done
