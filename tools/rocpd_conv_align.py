"""Per-layer comparison of the YOLO conv kernels: the i-th conv dispatch of a forward is the same layer in every run, so
kernel traces of the same probe taken with different TSTAR_YOLO_SW / TSTAR_YOLO_SW_P settings align by dispatch index.

    python tools/rocpd_conv_align.py <n_forwards> <label>=<db> <label>=<db> ...
"""
import re
import sqlite3
import sys

nfwd = int(sys.argv[1])
runs = []
for spec in sys.argv[2:]:
    label, path = spec.split("=", 1)
    c = sqlite3.connect(path)
    rows = [(n, e - s, gx) for n, s, e, gx in c.execute("select name, start, end, grid_x from kernels order by start") if "conv_" in n]
    per = len(rows) // nfwd
    rows = rows[len(rows) - per * (nfwd - 1):]                    # drop the first (warm-up) forward
    ops = []
    for i in range(per):
        sel = rows[i::per]
        name = re.sub(r"\(.*$", "", sel[0][0]).replace("void ", "").replace("tstar::", "")
        ops.append((name, sel[0][2] // 256 if sel[0][2] % 256 == 0 else sel[0][2], sum(d for _, d, _ in sel) / len(sel) / 1e3))
    runs.append((label, ops))
n = min(len(o) for _, o in runs)
print("| op | " + " | ".join(f"{l}: kernel, blocks, us" for l, _ in runs) + " | best |")
print("|---|" + "---|" * (len(runs) + 1))
tot = [0.0] * len(runs)
best_tot = 0.0
for i in range(n):
    cells, ts = [], []
    for r, (_, ops) in enumerate(runs):
        k, g, us = ops[i]
        cells.append(f"{k} {g} {us:.0f}")
        ts.append(us)
        tot[r] += us
    b = min(range(len(ts)), key=lambda r: ts[r])
    best_tot += ts[b]
    print(f"| {i} | " + " | ".join(cells) + f" | {runs[b][0]} |")
print("totals us: " + ", ".join(f"{l} {t:.0f}" for (l, _), t in zip(runs, tot)) + f", per-op best {best_tot:.0f}")
