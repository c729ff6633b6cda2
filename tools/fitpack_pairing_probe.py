"""Times the 63 fits of a reference-default search through csrc/fitpack.cpp with and without the paired steady loop (round 5) and checks
that both give the same bits.  CPU only: python tools/fitpack_pairing_probe.py"""
import sys, time, ctypes as C, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tstar_amd import spline_worker as SW
from test_host_logic import _reference_default_fit_problems
probs = _reference_default_fit_problems()
lib = SW._native_fit()
lib.tstar_curfit_pairing.restype = None
def run(pair, lanes):
    lib.tstar_curfit_pairing(pair)
    outs = []; tot = 0.0; per = []
    for (x, y) in probs:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = SW.curfit(x, y, 0.5, lanes); best = min(best, time.perf_counter() - t0)
        outs.append(r); tot += best; per.append(best)
    return outs, tot, per
for lanes in (8, 4):
    a, ta, pa = run(0, lanes); b, tb, pb = run(1, lanes)
    same = all(np.array_equal(u[0], v[0]) and np.array_equal(u[1], v[1]) and u[3] == v[3] for u, v in zip(a, b))
    print(f"lanes {lanes}: unpaired {ta*1e3:.1f} ms  paired {tb*1e3:.1f} ms  ({ta/tb:.2f}x)  bit-identical {same}; last fit {pa[-1]*1e3:.2f} -> {pb[-1]*1e3:.2f} ms")
