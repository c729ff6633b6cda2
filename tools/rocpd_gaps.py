"""GPU idle time between kernels in a rocprofv3 kernel trace (single stream): where the host is the bottleneck.

    python tools/rocpd_gaps.py <db> [fraction of the trace to keep, default 0.6]
    python tools/rocpd_gaps.py <db> --window-ms 1569      # the last 1569 ms (= the bench's timed region)
"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("tstar::", "")[:60]


c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# keep the last `frac` of the trace (the timed searches), default 60 %, or an explicit window
t0, t1 = rows[0][1], max(r[2] for r in rows)
if len(sys.argv) > 3 and sys.argv[2] == "--window-ms":
    cut = t1 - float(sys.argv[3]) * 1e6
else:
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[1] >= cut]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    if s1 > e0:
        gaps.append((s1 - e0, short(n0), short(n1)))
tot_gap = sum(g[0] for g in gaps)
print(f"window {span/1e6:.1f} ms, kernels busy {busy/1e6:.1f} ms ({100*busy/span:.1f} %), idle {tot_gap/1e6:.1f} ms in {len(gaps)} gaps")
for lo, hi in [(0, 5e3), (5e3, 20e3), (20e3, 100e3), (100e3, 1e6), (1e6, 1e12)]:
    sel = [g for g in gaps if lo <= g[0] < hi]
    print(f"  gaps {lo/1e3:>6.0f}-{hi/1e3:<8.0f} us: n={len(sel):5d} total {sum(g[0] for g in sel)/1e6:8.2f} ms")
print("largest gaps (us, after kernel -> before kernel):")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"  {g[0]/1e3:9.1f}  {g[1]}  ->  {g[2]}")
