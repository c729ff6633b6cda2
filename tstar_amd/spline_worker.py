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
import struct
import sys


def spline_distribution(visited_indices, observed_scores, video_length: int, s: float = 0.5):
    """spline_keyframe_distribution after the visited frames have been extracted
    (/root/reference/TStar/interface_searcher.py:262-274): uniform when nothing was visited; else the smoothing
    spline evaluated on every frame (extrapolating), floored at 1/N, squashed by a sigmoid and normalised.  The
    same numpy / scipy calls as the reference, on the host: P is bit-identical to the reference's on the same
    machine by construction (a device exp() could differ from numpy's in the last place)."""
    import numpy as np
    from scipy.interpolate import UnivariateSpline
    if len(visited_indices) == 0:
        return np.ones(video_length) / video_length
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
    from scipy.interpolate import UnivariateSpline
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
                t, c, k = UnivariateSpline(x, y, s=s)._eval_args
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
