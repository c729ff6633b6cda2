"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.

    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/r01_kernel_stats.md
    python tools/rocpd_stats.py <db> --window-ms 1569     # only the last 1569 ms of the trace (= the bench's
                                                          # timed region: ms_per_step x steps of the traced run)

Ends with the aggregate over every gemm_f32* dispatch (the figure to compare with bench.py's
roofline.avg_launch_ms, which is measured with HIP events inside the timed region).
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("tstar::", "")
    return name[:90]


def main(path, window_ms=None):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels").fetchall()
    if window_ms is not None:
        t_end = max(e for _, _, e in rows)
        rows = [r for r in rows if r[1] >= t_end - window_ms * 1e6]
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += (e - s)
    total = sum(v[1] for v in agg.values())
    print(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]}"
          + (f", last {window_ms:.0f} ms = the bench's timed region" if window_ms is not None else "") + ")\n")
    print(f"total kernel time {total/1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} | {100*t/total:.2f} |")


    gn = sum(n for name, (n, t) in agg.items() if "gemm_f32" in name)
    gt = sum(t for name, (n, t) in agg.items() if "gemm_f32" in name)
    if gn:
        print(f"\nall `gemm_f32*` dispatches: {gn} calls, {gt / 1e6:.3f} ms, average {gt / gn / 1e6:.4f} ms "
              f"({100 * gt / total:.1f} % of the kernel time)")


if __name__ == "__main__":
    w = None
    if "--window-ms" in sys.argv:
        i = sys.argv.index("--window-ms")
        w = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    main(sys.argv[1], w)
