"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("tstar::", "")
    return name[:90]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += (e - s)
    total = sum(v[1] for v in agg.values())
    print(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]})\n")
    print(f"total kernel time {total/1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} | {100*t/total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
