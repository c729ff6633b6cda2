"""Matrix-pipe utilisation per kernel family from one rocprofv3 PMC pass
(`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, rocpd sqlite output).

    python tools/rocpd_mfma.py <results.db> > profiles/rNN_pmc_mfma_utilisation.md

SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip (MI355X guide: = pipe cycles x wave-level
MFMA instructions, e.g. 64 per v_mfma_f32_32x32x2_f32), so
    utilisation = MFMA_BUSY / (1024 SIMDs x kernel cycles)
with kernel cycles taken two ways: GRBM_GUI_ACTIVE of the dispatch (both its raw value and /8 in case the
counter is summed over the 8 XCDs -- the plausible one is the one that keeps utilisation <= 1) and
duration x 2.4 GHz.
"""
import re
import sqlite3
import sys

SIMDS = 256 * 4
CLOCK_GHZ = 2.4


def family(name):
    n = re.sub(r"\(.*$", "", name).replace("void ", "").replace("tstar::", "")
    for f in ("gemm_f32_hybrid_kernel", "gemm_f32_kernel", "gemm_bf16w2_wide_kernel", "attention_f32_kernel", "ax3::attention_x3_kernel",
              "attention_split_kernel"):
        if n.startswith(f):
            return "gemm_f32* / gemm_bf16w2_wide_kernel (every launch through gemm_f32())" if f.startswith("gemm") else f
    return None


def main(path):
    c = sqlite3.connect(path)
    per = {}
    for name, cn, v, dur, did in c.execute("select name, counter_name, counter_value, duration, dispatch_id from pmc_events"):
        f = family(name)
        if f is None:
            continue
        d = per.setdefault((f, did), {"dur": dur})
        d[cn] = d.get(cn, 0.0) + v
    agg = {}
    for (f, _), d in per.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        a = agg.setdefault(f, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
        a[2] += d["GRBM_GUI_ACTIVE"]
        a[3] += d["dur"]
    print("# Matrix-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))\n")
    print("| kernel family | dispatches | MFMA busy cycles (sum) | GRBM_GUI_ACTIVE (sum) | duration ms (sum) | util vs GRBM | util vs GRBM/8 | util vs duration x 2.4 GHz |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for f, (n, busy, grbm, dur) in sorted(agg.items()):
        print(f"| `{f}` | {n} | {busy:.4g} | {grbm:.4g} | {dur / 1e6:.2f} | {busy / (SIMDS * grbm):.3f} | "
              f"{busy / (SIMDS * grbm / 8):.3f} | {busy / (SIMDS * dur * CLOCK_GHZ):.3f} |")


if __name__ == "__main__":
    main(sys.argv[1])
