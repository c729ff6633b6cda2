"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.

    python tools/rocpd_stats.py <db>                                   # the whole trace
    python tools/rocpd_stats.py <db> --timed-region [bench.json]       # only the bench's timed region: cut at the marker
                                                                       # kernels bench.py enqueues around it (tools/rocpd_window.py;
                                                                       # the JSON's host stamps are the fallback)
    python tools/rocpd_stats.py <db> --timed-region bench.json --check # exit 1 unless the trace reproduces the JSON's roofline leg:
                                                                       # |trace avg over the launches the HIP events sampled -
                                                                       # roofline.avg_launch_ms| <= 3 % and the number of
                                                                       # dominant-kernel dispatches = stride x launches_timed +- 2 %

Ends with the aggregate over every dispatch of the dominant kernel family (gemm_f32* / gemm_bf16w2_wide_kernel = every launch that goes through gemm_f32(), or conv_* for the YOLO-World
backend): the figure to compare with bench.py's roofline.avg_launch_ms, which is measured with HIP events inside the
timed region.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_window import kernel_rows  # noqa: E402


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("tstar::", "")
    return name[:90]


def main(path, timed=False, bench_json=None, check=False):
    rows, how, (lo, hi) = kernel_rows(path, bench_json, window_on=timed)
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += (e - s)
    total = sum(v[1] for v in agg.values())
    print(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]}"
          + (f", the bench's timed region: window from {how}, {(hi - lo) / 1e6:.1f} ms" if timed and lo is not None else
             (", " + how if timed else "")) + ")\n")
    print(f"total kernel time {total/1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} | {100*t/total:.2f} |")

    bj = json.load(open(bench_json)) if bench_json and os.path.isfile(bench_json) else None
    fam = ("conv_valu_kernel", "conv_sw_kernel", "conv_halo_kernel") if bj and bj.get("roofline", {}).get("bound") == "valu" else ("gemm_f32", "gemm_bf16w2_wide_kernel")
    gn = sum(n for name, (n, t) in agg.items() if any(f in name for f in fam))
    gt = sum(t for name, (n, t) in agg.items() if any(f in name for f in fam))
    ok = True
    if gn:
        print(f"\nall `{'* / '.join(fam)}*` dispatches: {gn} calls, {gt / 1e6:.3f} ms, average {gt / gn / 1e6:.4f} ms "
              f"({100 * gt / total:.1f} % of the kernel time)")
    if bj and timed:
        r = bj["roofline"]
        stride = r["timed_every_nth_launch"]
        want_n = r["launches_timed"] * stride
        avg_all = gt / max(gn, 1) / 1e6
        # the HIP events time ONE launch out of every `stride` consecutive ones, at a position given by a hash of the block
        # index (csrc/prof.hip: prof_start); the launches of the family are in stream order in the trace, so the same
        # subset can be cut out of it -- like for like, free of the sampling error of a 1-in-`stride` sample of launches
        # whose durations span 13 us .. 4 ms (186 samples of 930: +-5 % on the mean)
        fam_rows = sorted(((s_, e_) for name, s_, e_ in rows if any(f in name for f in fam)))
        sub = [e_ - s_ for idx, (s_, e_) in enumerate(fam_rows)
               if idx % stride == ((((idx // stride) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF) >> 33) % stride]
        avg = sum(sub) / max(len(sub), 1) / 1e6
        print(f"\nthe launches the HIP events sampled (1 of every {stride}, csrc/prof.hip's rule applied to the trace order): "
              f"{len(sub)} dispatches, average {avg:.4f} ms; all {gn} dispatches: {avg_all:.4f} ms "
              f"(the sample is {100 * (avg - avg_all) / avg_all:+.2f} % from the population)")
        d_avg = abs(avg - r["avg_launch_ms"]) / r["avg_launch_ms"]
        d_n = abs(gn - want_n) / max(want_n, 1)
        steps = bj["steps"]
        print(f"\nagainst the bench line of the same run: roofline.avg_launch_ms {r['avg_launch_ms']:.4f} (HIP events, every "
              f"{r['timed_every_nth_launch']}th launch, {r['launches_timed']} sampled) vs {avg:.4f} from the trace over the same launches: {100 * d_avg:.2f} % apart; "
              f"dispatches {gn} vs {r['timed_every_nth_launch']} x {r['launches_timed']} = {want_n}: {100 * d_n:.2f} % apart; "
              f"{gn / steps:.1f} dispatches per step; window {(hi - lo) / 1e6 if lo is not None else float('nan'):.1f} ms vs ms_per_step x steps = "
              f"{bj['ms_per_step'] * steps:.1f} ms")
        ok = d_avg <= 0.03 and d_n <= 0.02
        print("CHECK " + ("passed" if ok else "FAILED") + " (|avg| <= 3 %, dispatch count <= 2 %)")
    if check and not ok:
        sys.exit(1)


if __name__ == "__main__":
    a = sys.argv[1:]
    check = "--check" in a
    if check:
        a.remove("--check")
    timed = "--timed-region" in a
    bj = None
    if timed:
        i = a.index("--timed-region")
        if i + 1 < len(a) and not a[i + 1].endswith(".db"):
            bj = a[i + 1]
            del a[i + 1]
        del a[i]
    main(a[0], timed, bj, check)
