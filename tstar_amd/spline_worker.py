"""Worker process of tstar_amd.spline_pool: FITPACK smoothing-spline fits (and the sampling distribution built
from them) over a pipe.

Run as a SCRIPT (``python spline_worker.py``): it needs numpy and scipy only, so a worker starts in a fraction
of a second and never touches the GPU runtime.  The package imports ``spline_distribution`` from here so that the
in-process path and the workers run the same statements.

Protocol (little endian), one request at a time on stdin, one reply on stdout:
  request:  int64 m, float64 s, int64 N, float64 x[m], float64 y[m]
  reply ok, N == 0: int64 0, int64 16n, int64 k, float64 t[n], float64 c[n]     (c zero-padded to n)
  reply ok, N  > 0: int64 0, int64 8N, int64 -1, float64 P[N]   (spline_keyframe_distribution over N frames)
  reply err: int64 1, int64 len, int64 0, utf-8 message[len]
The fit is exactly the call the reference makes -- ``UnivariateSpline(x, y, s=s)``
(/root/reference/TStar/interface_searcher.py:265) -- through the same scipy, so (t, c, k) and P are bit-identical
to an in-process evaluation.
"""
import os
import struct
import sys

_FIT = None          # ctypes handle of libtstar_fitpack.so, False = unavailable / disabled


def _native_fit():
    """tstar_amd/libtstar_fitpack.so (csrc/fitpack.cpp: FITPACK's curfit restated operation for operation, its O(n^2)
    smoothing-parameter step run as a skewed SIMD pipeline; (t, c, fp) bit-identical to scipy's, 4-10x faster on late
    fits), or None -- then the same scipy call the reference makes is used.  TSTAR_NATIVE_FIT=0 disables it."""
    global _FIT
    if _FIT is None:
        _FIT = False
        if os.environ.get("TSTAR_NATIVE_FIT", "1") != "0":
            import ctypes as C
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtstar_fitpack.so")
            try:
                lib = C.CDLL(path)
                lib.tstar_curfit.restype = C.c_int
                lib.tstar_curfit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
                _FIT = lib
            except OSError:
                pass
    return _FIT or None


def curfit(x, y, s: float = 0.5, lanes: int = 0):
    """UnivariateSpline(x, y, s=s)._eval_args through the native restatement: (t [n], c [n] (the last 4 entries unused, zero),
    k = 3, fp, ier), or None when the library is absent, the input is outside what it restates (fewer than 4 points, not
    strictly increasing) or FITPACK reports a warning (ier > 0: scipy must raise it the way the reference sees it)."""
    lib = _native_fit()
    if lib is None:
        return None
    import ctypes as C
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    m = len(x)
    if m < 4 or x.shape != y.shape or x.ndim != 1 or not (np.isfinite(x).all() and np.isfinite(y).all()):
        return None
    t = np.zeros(m + 4)
    c = np.zeros(m + 4)
    n, fp, it = C.c_int(0), C.c_double(0.0), C.c_int(0)
    ier = lib.tstar_curfit(x.ctypes.data, y.ctypes.data, m, float(s), int(lanes), t.ctypes.data, c.ctypes.data, C.byref(n),
                           C.byref(fp), C.byref(it))
    if ier not in (0, -1, -2):
        return None
    return t[:n.value], c[:n.value], 3, fp.value, ier


def fit_tck(x, y, s: float = 0.5):
    """``UnivariateSpline(x, y, s=s)._eval_args`` = (t, c, k): through the native restatement when it applies (bit-identical), else
    scipy's own call -- what a worker answers to a plain fit request, and what the in-process fallback of spline_pool.fit_many runs."""
    fit = curfit(x, y, s)
    if fit is not None:
        return fit[:3]
    from scipy.interpolate import UnivariateSpline
    return UnivariateSpline(x, y, s=s)._eval_args


def spline_distribution(visited_indices, observed_scores, video_length: int, s: float = 0.5):
    """spline_keyframe_distribution after the visited frames have been extracted
    (/root/reference/TStar/interface_searcher.py:262-274): uniform when nothing was visited; else the smoothing
    spline evaluated on every frame (extrapolating), floored at 1/N, squashed by a sigmoid and normalised.  The
    same numpy / scipy calls as the reference, on the host: P is bit-identical to the reference's on the same
    machine by construction (a device exp() could differ from numpy's in the last place)."""
    import numpy as np
    if len(visited_indices) == 0:
        return np.ones(video_length) / video_length
    fit = curfit(visited_indices, observed_scores, s)
    if fit is not None:
        # the same (t, c, k) scipy's fit returns, bit for bit (tests/test_host_logic.py); evaluated by the same scipy splev
        # that UnivariateSpline.__call__ runs
        from scipy.interpolate import splev
        spline_scores = splev(np.arange(video_length), fit[:3])
    else:
        from scipy.interpolate import UnivariateSpline
        spline = UnivariateSpline(visited_indices, observed_scores, s=s)
        spline_scores = spline(np.arange(video_length))
    adjusted = np.maximum(1 / video_length, spline_scores)
    p = 1 / (1 + np.exp(-adjusted))
    p /= p.sum()
    return p


def _read(f, n):
    b = f.read(n)
    if len(b) != n:
        raise EOFError
    return b


def main():
    import numpy as np
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        hdr = inp.read(24)
        if len(hdr) < 24:
            return
        m, s, N = struct.unpack("<qdq", hdr)
        try:
            x = np.frombuffer(_read(inp, 8 * m), dtype=np.float64)
            y = np.frombuffer(_read(inp, 8 * m), dtype=np.float64)
        except EOFError:
            return
        try:
            if N > 0:
                P = np.ascontiguousarray(spline_distribution(x, y, int(N), s), dtype=np.float64)
                out.write(struct.pack("<qqq", 0, 8 * len(P), -1) + P.tobytes())
            else:
                t, c, k = fit_tck(x, y, s)
                t = np.ascontiguousarray(t, dtype=np.float64)
                cc = np.zeros(len(t), dtype=np.float64)
                cc[:len(c)] = c
                out.write(struct.pack("<qqq", 0, 16 * len(t), int(k)) + t.tobytes() + cc.tobytes())
        except Exception as e:                      # report, keep serving
            msg = f"{type(e).__name__}: {e}".encode()
            out.write(struct.pack("<qqq", 1, len(msg), 0) + msg)
        out.flush()


if __name__ == "__main__":
    main()
