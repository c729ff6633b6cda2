"""Worker process of tstar_amd.spline_pool: FITPACK smoothing-spline fits over a pipe.

Run as a SCRIPT (``python spline_worker.py``), never imported by the package: it needs numpy and scipy
only, so a worker starts in a fraction of a second and never touches the GPU runtime.

Protocol (little endian), one request at a time on stdin, one reply on stdout:
  request:  int64 m, float64 s, float64 x[m], float64 y[m]
  reply ok: int64 0, int64 n, int64 k, float64 t[n], float64 c[n]     (c zero-padded to n)
  reply err: int64 1, int64 len, int64 0, utf-8 message[len]
The fit is exactly the call the reference makes -- ``UnivariateSpline(x, y, s=s)``
(/root/reference/TStar/interface_searcher.py:265) -- through the same scipy, so (t, c, k) is bit-identical
to an in-process fit.
"""
import struct
import sys


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
        hdr = inp.read(16)
        if len(hdr) < 16:
            return
        m, s = struct.unpack("<qd", hdr)
        try:
            x = np.frombuffer(_read(inp, 8 * m), dtype=np.float64)
            y = np.frombuffer(_read(inp, 8 * m), dtype=np.float64)
        except EOFError:
            return
        try:
            t, c, k = UnivariateSpline(x, y, s=s)._eval_args
            t = np.ascontiguousarray(t, dtype=np.float64)
            cc = np.zeros(len(t), dtype=np.float64)
            cc[:len(c)] = c
            out.write(struct.pack("<qqq", 0, len(t), int(k)) + t.tobytes() + cc.tobytes())
        except Exception as e:                      # report, keep serving
            msg = f"{type(e).__name__}: {e}".encode()
            out.write(struct.pack("<qqq", 1, len(msg), 0) + msg)
        out.flush()


if __name__ == "__main__":
    main()
