"""Kernel time grouped by (kernel, grid size): shows which GEMM shapes dominate."""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("tstar::", "")[:70]


c = sqlite3.connect(sys.argv[1])
agg = {}
for name, gx, gy, wx, wy, s, e in c.execute("select name, grid_x, grid_y, workgroup_x, workgroup_y, start, end from kernels"):
    nb = gx // max(wx, 1)
    if gy and gy // max(wy, 1) > 1:
        nb = f"{nb}x{gy // max(wy, 1)}"
    a = agg.setdefault((name, nb), [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
print("| kernel | blocks | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|---:|")
for (name, nb), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"| `{short(name)}` | {nb} | {n} | {t / 1e6:.3f} | {t / n / 1e3:.1f} | {100 * t / tot:.2f} |")
