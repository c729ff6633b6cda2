"""Per-kernel PMC averages from a rocprofv3 (rocpd sqlite) counter-collection run.

    python tools/rocpd_pmc.py <results.db> [more.db ...]
"""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("tstar::", "")[:80]


def main(paths):
    for path in paths:
        c = sqlite3.connect(path)
        rows = c.execute("select name, counter_name, counter_value, duration, dispatch_id from pmc_events").fetchall()
        agg = {}
        for name, cn, v, dur, did in rows:
            a = agg.setdefault((name, cn), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += v
            a[2] += dur
        print(f"# {path.split('/')[-2] if '/' in path else path}: per-kernel counter totals / per-dispatch averages\n")
        print("| kernel | counter | dispatches | sum | avg per dispatch | avg duration us |")
        print("|---|---|---:|---:|---:|---:|")
        for (name, cn), (n, s, d) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
            print(f"| `{short(name)}` | {cn} | {n} | {s:.4g} | {s / n:.4g} | {d / n / 1e3:.1f} |")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
