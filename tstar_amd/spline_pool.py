"""Process pool for the host-side FITPACK fits of several searches advancing in lock-step.

``update_frame_distribution`` fits ``UnivariateSpline(frames, scores, s=0.5)`` once per search iteration
(/root/reference/TStar/interface_searcher.py:265).  The fit is sequential Fortran that holds the GIL and,
with the reference's default 4x4 grid (63 iterations per video), costs ~15-20 ms per call on 3600 frames --
more host time than the GPU needs for the iteration's detector work.  Within one search it overlaps with the
verification batch; across the items of a lock-step group (tstar_amd.lockstep) the fits are independent, so
they run here in worker processes, one item per worker, through the same scipy (bit-identical results).

Workers are plain ``python spline_worker.py`` subprocesses (numpy + scipy only, no GPU runtime, no fork of the
HIP-initialised parent) talking over pipes; they exit when the parent's pipe closes.
"""
from __future__ import annotations

import atexit
import os
import threading
import struct
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple

import numpy as np

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spline_worker.py")


class SplinePoolError(RuntimeError):
    pass


def host_threads() -> int:
    """Hardware threads this process may run on (its affinity mask / cgroup cpuset), not the machine's count: under a
    container or a per-rank cpuset ``os.cpu_count()`` over-reports and 8 ranks would oversubscribe the host."""
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        return max(1, os.cpu_count() or 1)


def cgroup_cpu_quota() -> Optional[float]:
    """CPUs' worth of time the cgroup allows this process (cpu.max of cgroup v2, cfs quota of v1), or None when unlimited / unknown:
    a quota without a cpuset does not show in the affinity mask."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            return float(q) / float(per)
    except (OSError, ValueError, IndexError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        if q > 0 and per > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def default_workers() -> int:
    """TSTAR_SPLINE_WORKERS, else up to 16 workers out of half this rank's share of the host threads it may use.  The share is the
    CONSERVATIVE split: (affinity mask, capped by the cgroup's CPU quota) // LOCAL_WORLD_SIZE -- whether the mask is the whole host, a
    container cpuset all local ranks share, or a per-rank cpuset cannot be told from inside one rank (a narrow shared cpuset looks
    like a per-rank one), and taking a shared mask for a private one oversubscribes the host 8-fold, while the opposite mistake
    only leaves workers unused.  A launcher that pins each rank to its own cpuset says so with TSTAR_SPLINE_MASK_PER_RANK=1 (the
    mask is then the share), or sets TSTAR_SPLINE_WORKERS outright."""
    env = os.environ.get("TSTAR_SPLINE_WORKERS")
    if env is not None:
        return max(0, int(env))
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    usable = host_threads()
    quota = cgroup_cpu_quota()
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    share = usable if os.environ.get("TSTAR_SPLINE_MASK_PER_RANK") == "1" else usable // ranks
    return max(1, min(16, share // 2))


class SplinePool:
    def __init__(self, workers: int):
        if workers < 1:
            raise ValueError("SplinePool needs at least one worker")
        self._procs: List[subprocess.Popen] = []
        self._lock = threading.Lock()           # one caller at a time owns the pipes (bench.py --concurrency)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        try:
            for _ in range(workers):
                self._procs.append(subprocess.Popen([sys.executable, _WORKER], stdin=subprocess.PIPE,
                                                    stdout=subprocess.PIPE, env=env))
        except OSError as e:
            self.close()
            raise SplinePoolError(f"cannot start spline workers: {e}") from e
        atexit.register(self.close)

    def __len__(self):
        return len(self._procs)

    def cpu_seconds(self) -> float:
        """user + system CPU seconds the worker processes have consumed so far (/proc/<pid>/stat fields 14, 15)."""
        tck = float(os.sysconf("SC_CLK_TCK"))
        total = 0.0
        for p in self._procs:
            try:
                with open(f"/proc/{p.pid}/stat") as f:
                    parts = f.read().rsplit(")", 1)[1].split()
                total += (int(parts[11]) + int(parts[12])) / tck
            except (OSError, IndexError, ValueError):
                pass
        return total

    def close(self):
        for p in self._procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self._procs:
            try:
                p.wait(timeout=2)
            except Exception:
                p.kill()
        self._procs = []

    @staticmethod
    def _checked(problems):
        """Every (x, y) as contiguous float64 1-D arrays of equal length -- raised BEFORE a byte goes to a worker: a caller's
        mistake must not cost the pool (a half-served pool cannot be resynchronised and is closed)."""
        out = []
        for x, y in problems:
            x = np.ascontiguousarray(x, dtype=np.float64)
            y = np.ascontiguousarray(y, dtype=np.float64)
            if x.shape != y.shape or x.ndim != 1:
                raise ValueError("spline fit needs two 1-D arrays of equal length")
            out.append((x, y))
        return out

    @staticmethod
    def _send(p: subprocess.Popen, x: np.ndarray, y: np.ndarray, s: float, n_frames: int = 0):
        p.stdin.write(struct.pack("<qdq", len(x), float(s), int(n_frames)) + x.tobytes() + y.tobytes())
        p.stdin.flush()

    @staticmethod
    def _recv(p: subprocess.Popen):
        hdr = p.stdout.read(24)
        if len(hdr) != 24:
            raise SplinePoolError("spline worker exited")
        status, nbytes, k = struct.unpack("<qqq", hdr)
        if status != 0:
            raise SplinePoolError("spline worker: " + p.stdout.read(nbytes).decode(errors="replace"))
        buf = p.stdout.read(nbytes)
        if len(buf) != nbytes:
            raise SplinePoolError("spline worker exited mid-reply")
        if k < 0:                                   # a distribution P[N]
            return np.frombuffer(buf, dtype=np.float64).copy()
        n = nbytes // 16
        t = np.frombuffer(buf, dtype=np.float64, count=n).copy()
        c = np.frombuffer(buf, dtype=np.float64, count=n, offset=8 * n).copy()
        return t, c, int(k)

    def fit_many(self, problems: Sequence[Tuple[np.ndarray, np.ndarray]], s: float = 0.5, n_frames=0):
        """[(x, y)] -> [(t, c, k)] with c zero-padded to len(t) (FITPACK's own layout), input order; with
        ``n_frames`` > 0 (one int, or one per problem) -> [P float64 [n_frames]]
        (spline_worker.spline_distribution) instead."""
        nf = list(n_frames) if hasattr(n_frames, "__len__") else [int(n_frames)] * len(problems)
        if len(nf) != len(problems):
            raise ValueError("fit_many: one n_frames per problem")
        if not self._procs:
            raise SplinePoolError("spline pool is closed")
        problems = self._checked(problems)
        out: List[Optional[Tuple[np.ndarray, np.ndarray, int]]] = [None] * len(problems)
        with self._lock:
            if not self._procs:
                raise SplinePoolError("spline pool is closed")
            w = len(self._procs)
            try:
                for lo in range(0, len(problems), w):
                    chunk = problems[lo:lo + w]
                    for j, (p, (x, y)) in enumerate(zip(self._procs, chunk)):
                        self._send(p, x, y, s, nf[lo + j])
                    for j, p in enumerate(self._procs[:len(chunk)]):
                        out[lo + j] = self._recv(p)
            except BaseException:
                self.close()                    # a half-served pool cannot be resynchronised
                raise
        return out


    def begin(self, problems: Sequence[Tuple[np.ndarray, np.ndarray]], s: float, n_frames: Sequence[int]) -> int:
        """Hand at most len(self) problems to the workers and return at once (the pool stays locked until ``end``)."""
        if len(problems) > len(self._procs) or len(n_frames) != len(problems):
            raise ValueError("SplinePool.begin: at most one problem per worker, one n_frames per problem")
        problems = self._checked(problems)
        self._lock.acquire()
        try:
            if not self._procs:
                raise SplinePoolError("spline pool is closed")
            for p, (x, y), n in zip(self._procs, problems, n_frames):
                self._send(p, x, y, s, n)
        except BaseException:
            self.close()
            self._lock.release()
            raise
        return len(problems)

    def end(self, count: int):
        """The replies of the ``begin`` before, in input order."""
        try:
            return [self._recv(p) for p in self._procs[:count]]
        except BaseException:                   # incl. KeyboardInterrupt / MemoryError: unread replies must never be
            self.close()                        # taken for the next call's
            raise
        finally:
            self._lock.release()


_pool: Optional[SplinePool] = None
_pool_failed = False


def get_pool() -> Optional[SplinePool]:
    """The process-wide pool (created on first use), or None when disabled (TSTAR_SPLINE_WORKERS=0) or
    when the workers could not be started (the caller then fits in-process with the same scipy call)."""
    global _pool, _pool_failed
    if _pool is not None and len(_pool) > 0:
        return _pool
    if _pool_failed:
        return None
    n = default_workers()
    if n == 0:
        return None
    try:
        _pool = SplinePool(n)
    except SplinePoolError as e:
        _pool_failed = True
        print(f"tstar_amd: {e}; fitting splines in-process", file=sys.stderr)
        return None
    return _pool


def _pooled(problems, s: float, n_frames):
    global _pool_failed
    if len(problems) > 1:
        pool = get_pool()
        if pool is not None:
            try:
                return pool.fit_many(problems, s, n_frames)
            except (OSError, SplinePoolError) as e:
                _pool_failed = True
                print(f"tstar_amd: spline pool failed ({e}); fitting in-process", file=sys.stderr)
    return None


def fit_many(problems: Sequence[Tuple[np.ndarray, np.ndarray]], s: float = 0.5):
    """Fit every (x, y) with ``UnivariateSpline(x, y, s=s)``: in the worker pool when there is more than one
    problem and a pool is available, otherwise in-process.  Returns [(t, c, k)] in input order."""
    out = _pooled(problems, s, 0)
    if out is not None:
        return out
    from .spline_worker import fit_tck         # the workers' own fit (native restatement when built, else scipy): same bits, same speed
    return [fit_tck(np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64), s) for x, y in problems]


def distribution_many(problems: Sequence[Tuple[np.ndarray, np.ndarray]], n_frames: Sequence[int], s: float = 0.5):
    """The sampling distribution P (float64 [n_frames[i]]) of every (visited frames, scores) problem --
    spline_worker.spline_distribution, i.e. interface_searcher.py:262-274 -- in the worker pool when there is
    more than one problem and a pool is available, otherwise in-process (same statements, same libraries)."""
    nf = [int(n) for n in n_frames]
    if len(nf) != len(problems) or any(n < 1 for n in nf):
        raise ValueError("distribution_many: one positive n_frames per problem")
    out = _pooled(problems, s, nf)
    if out is not None:
        return out
    from .spline_worker import spline_distribution
    return [spline_distribution(x, y, n, s) for (x, y), n in zip(problems, nf)]


def distribution_async(problems: Sequence[Tuple[np.ndarray, np.ndarray]], n_frames: Sequence[int], s: float = 0.5):
    """``distribution_many`` started NOW in the worker processes; returns ``wait() -> [P]``.  The caller enqueues GPU work
    (the iteration's verification batch) between the two, so the fit runs beside the enqueue instead of after it.  Also for a
    single problem (one search running alone: the fit is on its critical path).  Without a pool, or with more problems than
    workers, ``wait`` runs ``distribution_many`` -- the same statements, the same P."""
    global _pool_failed
    nf = [int(n) for n in n_frames]
    if len(nf) != len(problems) or any(n < 1 for n in nf):
        raise ValueError("distribution_async: one positive n_frames per problem")
    problems = [(np.array(x, dtype=np.float64), np.array(y, dtype=np.float64)) for x, y in problems]      # the caller's buffers may be reused
    pool = get_pool() if problems else None
    if pool is not None and len(problems) <= len(pool):
        try:
            count = pool.begin(problems, s, nf)
        except (OSError, SplinePoolError) as e:
            _pool_failed = True
            print(f"tstar_amd: spline pool failed ({e}); fitting in-process", file=sys.stderr)
        else:
            def wait():
                global _pool_failed
                try:
                    return pool.end(count)
                except (OSError, SplinePoolError) as e:
                    _pool_failed = True
                    print(f"tstar_amd: spline pool failed ({e}); fitting in-process", file=sys.stderr)
                    from .spline_worker import spline_distribution
                    return [spline_distribution(x, y, n, s) for (x, y), n in zip(problems, nf)]
            return wait
    return lambda: distribution_many(problems, nf, s)
