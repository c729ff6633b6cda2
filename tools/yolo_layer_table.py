"""Per-layer table of the YOLO-World convolutions from a kernel trace of tools/yolo_forward_probe.py:
layer shape, GFLOP, the kernel the launcher chose, workgroups, average microseconds, TFLOP/s, and totals by layer family.

    python tools/yolo_layer_table.py <db> <B> <n_forwards> [scale]
"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tstar_amd import yolo_world as Y

db, B, nfwd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scale = sys.argv[4] if len(sys.argv) > 4 else "l"
prog = Y.build_program(Y.synthetic_state_dict(0, scale), scale)
convs = []
for o in prog["ops"]:
    if o[0] != Y.OP_CONV:
        continue
    _, src, so, cin, dst, do, cout, ks, st = [int(v) for v in o[:9]]
    H, W, _ = prog["bufs"][src]
    Ho, Wo, _ = prog["bufs"][dst]
    convs.append((H, W, cin, cout, ks, st, Ho, Wo, 2.0 * B * Ho * Wo * cout * ks * ks * cin))
c = sqlite3.connect(db)
rows = [(n, e - s, gx, wx) for n, s, e, gx, wx in c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start") if "conv_" in n]
per = len(rows) // nfwd
assert per == len(convs), (per, len(convs))
rows = rows[len(rows) - per * (nfwd - 1):]                        # drop the first (warm-up) forward
print(f"# YOLO-World-v2-{scale.upper()} convolutions, B = {B}, launcher's own kernel choice ({nfwd - 1} forwards averaged)\n")
print("| op | input | cin -> cout | k / s | GFLOP | kernel | workgroups | us | TFLOP/s |")
print("|---:|---|---|---|---:|---|---:|---:|---:|")
fam = {}
tot_f = tot_t = 0.0
for i, (H, W, cin, cout, ks, st, Ho, Wo, fl) in enumerate(convs):
    sel = rows[i::per]
    us = sum(d for _, d, _, _ in sel) / len(sel) / 1e3
    name = re.sub(r"\(.*$", "", sel[0][0]).replace("void ", "").replace("tstar::", "")
    wgs = sel[0][2] // max(sel[0][3], 1)
    print(f"| {i} | {H}x{W} | {cin} -> {cout} | {ks} / {st} | {fl / 1e9:.1f} | `{name}` | {wgs} | {us:.0f} | {fl / us / 1e6:.1f} |")
    a = fam.setdefault((ks, st, Ho), [0.0, 0.0, 0])
    a[0] += fl; a[1] += us; a[2] += 1
    tot_f += fl; tot_t += us
print(f"\nall convolutions: {tot_f / 1e12:.3f} TFLOP in {tot_t / 1e3:.2f} ms = {tot_f / tot_t / 1e6:.1f} TFLOP/s\n")
print("| family (k, stride, output map) | layers | GFLOP | ms | TFLOP/s | share of conv time |")
print("|---|---:|---:|---:|---:|---:|")
for (ks, st, Ho), (f, u, n) in sorted(fam.items()):
    print(f"| {ks}x{ks} / {st} -> {Ho}x{Ho} | {n} | {f / 1e9:.0f} | {u / 1e3:.2f} | {f / u / 1e6:.1f} | {100 * u / tot_t:.1f} % |")
