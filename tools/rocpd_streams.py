"""Per-stream view of a rocprofv3 kernel trace: how busy each HIP stream / hardware queue is inside the timed region, how much of the time
two streams execute kernels at once, and which kernel families run on which stream (round 6: the speculative next-grid forward beside the
verification batch -- tstar_amd/lockstep.py).

    python tools/rocpd_streams.py <db> [--timed-region [bench.json]]
"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_window import window  # noqa: E402


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("tstar::", "")[:56]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


a = sys.argv[1:]
conn = sqlite3.connect(a[0])
cur = conn.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
key = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
print(f"# columns of `kernels`: {', '.join(cols)}\n# streams told apart by: {key}")
lo = hi = None
if "--timed-region" in a:
    i = a.index("--timed-region")
    lo, hi, how = window(conn, a[i + 1] if i + 1 < len(a) else None)
    print(f"# window: {how}")
rows = conn.execute(f"select name, start, end, {key or '0'} from kernels order by start").fetchall()
rows = [r for r in rows if "prof_mark_" not in r[0] and (lo is None or (r[1] >= lo and r[2] <= hi))]
span = max(r[2] for r in rows) - min(r[1] for r in rows)
by = {}
for n, s, e, k in rows:
    by.setdefault(k, []).append((n, s, e))
all_busy = union([(s, e) for _, s, e, _ in rows])
print(f"window {span / 1e6:.1f} ms; some kernel running {all_busy / 1e6:.1f} ms ({100 * all_busy / span:.1f} %); sum of kernel durations {sum(e - s for _, s, e, _ in rows) / 1e6:.1f} ms")
busy = {}
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for _, s, e in kv[1])):
    busy[k] = union([(s, e) for _, s, e in v])
    fam = {}
    for n, s, e in v:
        f = fam.setdefault(short(n), [0, 0])
        f[0] += 1
        f[1] += e - s
    top = sorted(fam.items(), key=lambda kv: -kv[1][1])[:6]
    print(f"\nstream {k}: {len(v)} dispatches, busy {busy[k] / 1e6:.1f} ms ({100 * busy[k] / span:.1f} % of the window)")
    for f, (c, t) in top:
        print(f"    {t / 1e6:8.2f} ms  n={c:5d}  {f}")
ks = sorted(busy, key=lambda k: -busy[k])
for i in range(len(ks)):
    for j in range(i + 1, len(ks)):
        a_, b_ = [(s, e) for _, s, e in by[ks[i]]], [(s, e) for _, s, e in by[ks[j]]]
        both = busy[ks[i]] + busy[ks[j]] - union(a_ + b_)
        if both > 0:
            print(f"\nstreams {ks[i]} and {ks[j]} execute kernels at the same time for {both / 1e6:.1f} ms "
                  f"({100 * both / max(busy[ks[j]], 1):.0f} % of stream {ks[j]}'s busy time)")
