"""Sharding of independent (video, question) items over the GPUs of one node.

The reference's dataset runner is a sequential loop in one process
(/root/reference/LVHaystackBench/run_TStar_onDataset.py:195-205).  Items are independent
(a new searcher per item, :108,125; the heuristic's per-item state is reset by
reparameterize_object_list), so the path shards with NO data-path collective: item i goes to
rank i % world, every rank seeds its sampler per item (seed = base + item id, so results do
not depend on the rank count), and ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU
tests) collects the K keyframe indices of every item at the end -- about 1 KB for 32 videos,
latency-bound.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading
from typing import Callable, List, Sequence

_COMM = None            # this process' tstar_comm handle (RCCL communicator created through the C ABI), or False = unavailable
_COMM_NOTE = ""         # why the library's communicator is not in use (reported in LAST_GATHER_PATH)
COMM_TIMED_OUT = False  # the watchdog gave up on tstar_comm_create: a thread of this process may still sit inside ncclCommInitRank
PREFER_RCCL = True      # gather through the library's own RCCL communicator whenever the ranks have GPUs (False / TSTAR_GATHER=gloo: never)
LAST_GATHER_PATH = None  # which way the last gather_keyframes() went (bench.py reports it as config.collective_path)


def comm_timeout_s() -> float:
    """Seconds the watchdog gives ``tstar_comm_create`` (ncclCommInitRank, a collective that can HANG rather than fail when a peer never
    arrives or the fabric bootstrap stalls) before the gather falls back to torch.distributed.  TSTAR_COMM_TIMEOUT_S, default 90."""
    try:
        return max(1.0, float(os.environ.get("TSTAR_COMM_TIMEOUT_S", "90")))
    except ValueError:
        return 90.0


def _create_with_watchdog(lib, uid: bytes, world: int, rank: int, timeout_s: float, create=None):
    """``tstar_comm_create`` on a helper thread (ctypes drops the GIL for the call) -> (handle or None, note).  On a timeout the thread is
    left behind as a daemon -- ncclCommInitRank cannot be cancelled -- and the caller must not touch the communicator it may still produce."""
    global COMM_TIMED_OUT
    h = C.c_void_p()
    box = {}
    dev = None
    try:
        import torch
        dev = torch.cuda.current_device()
    except Exception:
        pass

    def work():
        try:
            if dev is not None:
                import torch
                torch.cuda.set_device(dev)               # the current device is per thread: ncclCommInitRank binds to it
            box["rc"] = (create or lib.tstar_comm_create)(C.byref(h), uid, world, rank)
            if box["rc"] != 0:
                box["err"] = lib.tstar_last_error().decode()      # the library's error text is per thread: read it on this one
        except BaseException as e:                        # noqa: BLE001 -- reported to the caller's thread
            box["exc"] = e

    t = threading.Thread(target=work, name="tstar-comm-create", daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        COMM_TIMED_OUT = True
        return None, f"tstar_comm_create (ncclCommInitRank, {world} ranks) did not return within {timeout_s:.0f} s: watchdog fallback"
    if "exc" in box:
        return None, f"tstar_comm_create raised {box['exc']!r}"
    if box.get("rc") != 0:
        return None, box.get("err") or f"tstar_comm_create failed (code {box.get('rc')})"
    return h, ""


def _tstar_comm(world: int, rank: int, ctl):
    """The library's own RCCL communicator (include/tstar_hip.h: tstar_comm_*), bootstrapped over the already
    initialised torch.distributed group -- whatever its backend: the control messages (flags, the unique id) travel as ``ctl``-device
    tensors / pickled objects, so a gloo group is enough and no second RCCL communicator (torch's own) has to exist.  Rank 0 draws the
    unique id, the id is broadcast, every rank joins.  Returns the handle, or None when RCCL could not be bound or the communicator did
    not come up on every rank (ranks sharing one GPU: "Duplicate GPU detected"); the caller then gathers through torch.distributed."""
    global _COMM, _COMM_NOTE
    if _COMM is not None:
        return _COMM or None
    import torch
    import torch.distributed as dist
    from . import _lib
    lib = _lib.load()
    # ONE RCCL in the process: torch.distributed's "nccl" backend has already loaded the copy PyTorch ships; hand the library that very
    # file (dlopen of a loaded file returns the loaded object) instead of letting the bare name "librccl.so" resolve through the linker
    # cache to another installation's copy.  TSTAR_RCCL_LIB set by the user wins.
    if not os.environ.get("TSTAR_RCCL_LIB"):
        cand = os.path.join(os.path.dirname(os.path.abspath(torch.__file__)), "lib", "librccl.so")
        if os.path.isfile(cand):
            os.environ["TSTAR_RCCL_LIB"] = cand
    # every rank binds RCCL locally FIRST and the outcome is MIN-reduced: ncclCommInitRank is collective, so a rank that
    # cannot load the library must be known before any rank enters it (the others would block inside it forever)
    have = 1 if lib.tstar_comm_available() == 0 else 0
    if not have:
        _COMM_NOTE = lib.tstar_last_error().decode()
        print(f"tstar_amd[rank {rank}]: {_COMM_NOTE}; gathering through torch.distributed", file=sys.stderr)
    pre = torch.tensor([have], dtype=torch.int32, device=ctl)
    dist.all_reduce(pre, op=dist.ReduceOp.MIN)
    if int(pre.item()) != 1:
        _COMM_NOTE = _COMM_NOTE or "RCCL is not loadable on another rank"
        _COMM = False
        return None
    idbuf = C.create_string_buffer(128)
    box = [None]
    if rank == 0:
        box[0] = bytes(idbuf.raw) if lib.tstar_comm_unique_id(idbuf) == 0 else None
        if box[0] is None:
            print(f"tstar_amd: {lib.tstar_last_error().decode()}; gathering through torch.distributed", file=sys.stderr)
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        _COMM_NOTE = "ncclGetUniqueId failed on rank 0"
        _COMM = False
        return None
    # ncclCommInitRank under a watchdog: an ERROR comes back as a code, a HANG (a peer that never arrives, a stalled bootstrap) would
    # otherwise eat the whole job's time limit.  Every rank then agrees (MIN) on whether the communicator is up.
    h, note = _create_with_watchdog(lib, box[0], world, rank, comm_timeout_s())
    ok = h is not None
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if ok:
            lib.tstar_comm_destroy(h)
            note = "tstar_comm_create failed or timed out on another rank"
        _COMM_NOTE = note
        print(f"tstar_amd[rank {rank}]: {note}; gathering through torch.distributed", file=sys.stderr)
        _COMM = False
        return None
    _COMM = h
    return h


def close_comm():
    """Destroy the library's communicator (before torch.distributed.destroy_process_group)."""
    global _COMM, _COMM_NOTE
    if _COMM:
        from . import _lib
        _lib.load().tstar_comm_destroy(_COMM)
    _COMM = None
    _COMM_NOTE = ""



def shard_items(n_items: int, world: int, rank: int) -> List[int]:
    """Item ids of this rank (round robin; item cost is ~constant: the 1000-frame budget cap)."""
    return list(range(rank, n_items, world))


def item_seed(base_seed: int, item_id: int) -> int:
    return int(base_seed) + int(item_id)


def gather_keyframes(local_rows: Sequence[Sequence[int]], world: int, pad_to: int | None = None) -> List[List[int]]:
    """All-gather int32 [rows_per_rank, K] (padded with -1) -> list of rows in rank-major order.

    With world == 1 (or no initialised process group) this is the identity."""
    global LAST_GATHER_PATH
    rows = [list(map(int, r)) for r in local_rows]
    if world <= 1:
        LAST_GATHER_PATH = "identity (world 1: nothing to gather)"
        return rows
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("gather_keyframes: torch.distributed is not initialised")
    # control plane = the initialised torch.distributed group, whatever its backend (gloo: host tensors; nccl: device tensors); data
    # path = ONE ncclAllGather on the library's own RCCL communicator whenever the ranks have GPUs, else torch.distributed.all_gather
    ctl = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    want_rccl = PREFER_RCCL and torch.cuda.is_available() and os.environ.get("TSTAR_GATHER", "rccl") != "gloo"
    k = max([len(r) for r in rows], default=0)
    shape = torch.tensor([len(rows), k], dtype=torch.int64, device=ctl)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)
    nrow, k = int(shape[0]), int(shape[1])
    if pad_to is not None:
        nrow = max(nrow, pad_to)
    # wire format per rank: [row count, nrow * k indices padded with -1].  The explicit count keeps an item that found
    # ZERO keyframes in its place (an all-(-1) row) instead of shifting every later item of that rank.
    host = torch.full((1 + nrow * k,), -1, dtype=torch.int32)
    host[0] = len(rows)
    for i, r in enumerate(rows):
        if r:
            host[1 + i * k:1 + i * k + len(r)] = torch.tensor(r, dtype=torch.int32)
    n_i32 = 1 + nrow * k
    comm = _tstar_comm(world, dist.get_rank(), ctl) if want_rccl else None
    if comm is not None:                         # the C-ABI entry a non-Python host would call: ncclAllGather on our stream
        from . import _lib
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = host.to(dev)
        flat = torch.empty((world, n_i32), dtype=torch.int32, device=dev)
        _lib.check(_lib.load().tstar_allgather_i32(comm, buf.data_ptr(), flat.data_ptr(), n_i32, _lib.stream_ptr()),
                   "tstar_allgather_i32")
        out = list(flat.cpu())                   # synchronises the stream
        LAST_GATHER_PATH = (f"tstar_allgather_i32: ncclAllGather on the library's own RCCL communicator "
                            f"({world} ranks, {n_i32} int32 per rank, the caller's stream; control plane: torch.distributed over {dist.get_backend()})")
    else:
        buf = host.to(ctl)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        LAST_GATHER_PATH = (f"torch.distributed.all_gather over {dist.get_backend()} ({world} ranks)"
                            + (f"; the library's RCCL communicator was unavailable ({_COMM_NOTE})" if want_rccl else ""))
    res: List[List[int]] = []
    for t in out:
        flat_r = t.cpu().numpy()
        cnt = int(flat_r[0])
        if cnt < 0 or cnt > nrow:
            raise RuntimeError(f"gather_keyframes: a rank reported {cnt} rows of at most {nrow}")
        body = flat_r[1:].reshape(nrow, k) if k else None
        for i in range(cnt):
            res.append([int(v) for v in body[i] if v >= 0] if k else [])
    return res


def interleave_by_item(gathered: List[List[int]], n_items: int, world: int) -> List[List[int]]:
    """Undo the rank-major order of gather_keyframes for a round-robin sharding: result[i] = item i."""
    per_rank = [len(shard_items(n_items, world, r)) for r in range(world)]
    out: List[List[int]] = [[] for _ in range(n_items)]
    pos = 0
    for r in range(world):
        for j in range(per_rank[r]):
            out[r + j * world] = gathered[pos]
            pos += 1
    return out


def run_sharded(n_items: int, search_item: Callable[[int], Sequence[int]], world: int, rank: int, *,
                search_group: Callable[[List[int]], Sequence[Sequence[int]]] | None = None, group_size: int = 1) -> List[List[int]]:
    """Run this rank's items and gather all results, ordered by item id (identical on every rank).

    ``search_item(item_id) -> keyframe indices`` runs one item; with ``search_group`` (``[item ids] -> [keyframe indices
    per item]``, e.g. a ``tstar_amd.lockstep.search_lockstep`` group) this rank's items are handed over ``group_size`` at a
    time instead -- results must not depend on the grouping (per-item sampler seeds)."""
    ids = shard_items(n_items, world, rank)
    if search_group is not None and group_size > 1:
        mine = []
        for g0 in range(0, len(ids), group_size):
            grp = ids[g0:g0 + group_size]
            rows = [list(r) for r in search_group(grp)]
            if len(rows) != len(grp):
                raise ValueError("run_sharded: search_group must return one row per item")
            mine.extend(rows)
    else:
        mine = [list(search_item(i)) for i in ids]
    gathered = gather_keyframes(mine, world, pad_to=(n_items + world - 1) // world)
    if world <= 1:
        return gathered
    return interleave_by_item(gathered, n_items, world)
