"""Cut a rocprofv3 kernel trace (rocpd sqlite) to the bench's timed region.

bench.py brackets its timed region with two empty marker kernels (tstar_prof_mark: prof_mark_begin_kernel /
prof_mark_end_kernel, enqueued on the launch stream right after the opening barrier and right before the closing one),
so the window is taken on the GPU's own timeline: [end of the last begin marker, start of the last end marker].  The
bench JSON also carries host stamps of the same two instants (`config.timed_region`: CLOCK_REALTIME / MONOTONIC /
BOOTTIME ns) as a cross-check; they are only used when a trace has no markers (whichever clock lands inside the trace).

    from rocpd_window import kernel_rows
    rows, how = kernel_rows(db_path, bench_json_path_or_None)     # rows: (name, start, end[, extra columns])
"""
import json
import sqlite3

BEGIN, END = "prof_mark_begin_kernel", "prof_mark_end_kernel"


def window(conn, bench_json=None):
    """-> (t_lo, t_hi, description) in trace nanoseconds, or (None, None, reason) when the trace cannot be windowed."""
    marks = conn.execute("select name, start, end from kernels where name like '%prof_mark_%' order by start").fetchall()
    b = [r for r in marks if BEGIN in r[0]]
    e = [r for r in marks if END in r[0]]
    if b and e and e[-1][1] > b[-1][2]:
        return b[-1][2], e[-1][1], f"marker kernels ({len(b)} begin / {len(e)} end in the trace; the last pair is used)"
    if bench_json:
        try:
            tr = json.load(open(bench_json))["config"]["timed_region"]
            lo, hi = conn.execute("select min(start), max(end) from kernels").fetchone()
            for clock in ("boottime", "monotonic", "realtime"):
                t0, t1 = tr.get(f"t0_{clock}_ns"), tr.get(f"t1_{clock}_ns")
                if t0 and t1 and lo <= t0 <= hi and lo <= t1 <= hi + 2_000_000_000:
                    return t0, t1, f"host stamps of the timed region (CLOCK_{clock.upper()})"
        except (OSError, KeyError, ValueError):
            pass
    return None, None, "no marker kernels and no usable host stamps: whole trace"


def kernel_rows(db_path, bench_json=None, columns="name, start, end", window_on=True):
    conn = sqlite3.connect(db_path)
    lo, hi, how = window(conn, bench_json) if window_on else (None, None, "whole trace")
    rows = conn.execute(f"select {columns} from kernels order by start").fetchall()
    names = [c.strip() for c in columns.split(",")]
    i_s, i_e, i_n = names.index("start"), names.index("end"), names.index("name")
    rows = [r for r in rows if "prof_mark_" not in r[i_n]]
    if lo is not None:
        rows = [r for r in rows if r[i_s] >= lo and r[i_e] <= hi]
    return rows, how, (lo, hi)
