"""GPU idle time between kernels in a rocprofv3 kernel trace (single stream): where the host is the bottleneck.

    python tools/rocpd_gaps.py <db> [fraction of the trace to keep, default 0.6]
    python tools/rocpd_gaps.py <db> --timed-region [bench.json]     # the bench's timed region (marker kernels, tools/rocpd_window.py)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_window import kernel_rows  # noqa: E402


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("tstar::", "")[:60]


a = sys.argv[1:]
if "--timed-region" in a:
    i = a.index("--timed-region")
    bj = a[i + 1] if i + 1 < len(a) else None
    rows, how, _ = kernel_rows(a[0], bj)
    print(f"window: {how}")
else:
    rows, _, _ = kernel_rows(a[0], None, window_on=False)
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    frac = float(a[1]) if len(a) > 1 else 0.6
    cut = t1 - (t1 - t0) * frac
    rows = [r for r in rows if r[1] >= cut]
rows = sorted(rows, key=lambda r: r[1])
span = max(r[2] for r in rows) - rows[0][1]
# kernels of several streams may overlap: a gap is a stretch in which NO kernel runs (after the kernel that ended last)
gaps = []
last_n, last_e = rows[0][0], rows[0][2]
for n1, s1, e1 in rows[1:]:
    if s1 > last_e:
        gaps.append((s1 - last_e, short(last_n), short(n1)))
    if e1 > last_e:
        last_n, last_e = n1, e1
busy = span - sum(g[0] for g in gaps)
tot_gap = sum(g[0] for g in gaps)
print(f"window {span/1e6:.1f} ms, kernels busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), idle {tot_gap/1e6:.1f} ms in {len(gaps)} gaps")
for lo, hi in [(0, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)]:
    sel = [g for g in gaps if lo <= g[0] < hi]
    print(f"  gaps {lo/1e3:>6.0f}-{hi/1e3:<8.0f} us: n={len(sel):5d} total {sum(g[0] for g in sel)/1e6:8.2f} ms")
pairs = {}
for g in gaps:
    k = (g[1], g[2])
    n, t = pairs.get(k, (0, 0))
    pairs[k] = (n + 1, t + g[0])
print("idle by kernel pair (total ms, count, average us, after kernel -> before kernel):")
for k, (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"  {t/1e6:8.2f} ms  n={n:5d}  avg {t/n/1e3:8.1f}  {k[0]}  ->  {k[1]}")
print("largest gaps (us, after kernel -> before kernel):")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]/1e3:9.1f}  {g[1]}  ->  {g[2]}")
