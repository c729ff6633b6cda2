"""HBM-side traffic of one kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/rocpd_traffic.py <fetch.db> <write.db> <kernel substring[|substring...]> [command text] [bench line .json of the FETCH pass] > profiles/rNN_pmc_traffic.json

The fifth argument is the JSON line bench.py printed DURING the counter pass: its `config.population` (steps, lock-step shape, chunk
size, mode, grid ...) and `roofline.algorithmic_bytes_per_launch` are stored beside the counters, and bench.py only quotes the traffic
figure for a run whose population is the same (round 6: no more ratios across different launch populations).

Per the MI355X guide: both counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a
wide coalesced read, so it is doubled.  The counters sit on the L2's fabric side and include
Infinity-Cache hits.
"""
import json
import sqlite3
import sys


def total(path, counter, sub):
    c = sqlite3.connect(path)
    n = s = 0
    for name, cn, v in c.execute("select name, counter_name, counter_value from pmc_events"):
        if cn == counter and any(x in name for x in sub.split("|")):
            n += 1
            s += v
    return n, s


def main(fetch_db, write_db, sub, command=None, bench_json=None):
    nf, f = total(fetch_db, "FETCH_SIZE", sub)
    nw, w = total(write_db, "WRITE_SIZE", sub)
    out = {"kernel": sub, "fetch_dispatches": nf, "write_dispatches": nw,
           "FETCH_SIZE_KiB_sum": f, "WRITE_SIZE_KiB_sum": w,
           "bytes_per_launch_corrected": (2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
           "bytes_per_launch_uncorrected": (f / max(nf, 1) + w / max(nw, 1)) * 1024.0,
           "correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KiB units",
           "command": command or "rocprofv3 --kernel-trace --pmc <COUNTER> -- python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-grid4 --no-verify "
                      "(tools/collect_profiles.sh)"}
    if bench_json:
        try:
            line = [l for l in open(bench_json).read().splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            out["population"] = j["config"]["population"]
            out["algorithmic_bytes_per_launch"] = j["roofline"]["algorithmic_bytes_per_launch"]
            out["launches_total"] = j["roofline"]["launches_total"]
        except Exception as e:                                  # the counters are still worth keeping
            out["population_error"] = repr(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
