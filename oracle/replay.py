"""ORACLE (test infrastructure, never shipped, never on the product path).

Teacher-forced replay (parity level L2): the HIP pipeline runs closed-loop while a recorder
keeps what it produced batch by batch (cell confidences, class masks, dense scores); afterwards those numbers are
fed to the oracle searcher (oracle/searcher_ref.SearcherRef, the restatement of
/root/reference/TStar/interface_searcher.py:444-491 pinned by goldens G1-G6) in the order the reference's loop asks
for them.  Identical sampled seconds, histories and keyframes then prove the device searcher state machine, the
speculative verification batching and the host glue -- for exactly the confidences the GPU produced.

Used by tests/ and by bench.py's post-run keyframe check (as the checker, outside the timed region).
"""
from __future__ import annotations

import numpy as np

from . import searcher_ref as S


class Recorder:
    """Wraps ``heuristic.score_batch`` of a tstar_amd OWLInterface and records what the searcher consumed."""

    def __init__(self, h, keep_images: bool = True):
        self.h = h
        self.calls = []
        self._orig = h.score_batch

        def rec(d_images, rows, cols, image_sets=None, **kw):          # (lane=: the workspace of the forward, passed through)
            r = self._orig(d_images, rows, cols, image_sets=image_sets, **kw)
            self.calls.append(dict(_res=r, rows=rows, cols=cols, images=d_images.cpu().numpy() if keep_images else None,
                                   conf=r.cell_conf.cpu().numpy(), mask=r.cell_mask.cpu().numpy().astype(np.uint32),
                                   scores=r.scores.cpu().numpy(), boxes=r.boxes.cpu().numpy(), labels=r.labels.cpu().numpy()))
            return r
        h.score_batch = rec
        # a grid forward the searcher queued speculatively and then discarded (tstar_amd.lockstep._Group.speculate: the search ended
        # with the verification batch before it) is not something the searcher CONSUMED: drop its record
        h._speculation_dropped = lambda res: self.calls.__setitem__(slice(None), [c for c in self.calls if c["_res"] is not res])

    def restore(self):
        self.h.score_batch = self._orig
        if hasattr(self.h, "_speculation_dropped"):
            del self.h._speculation_dropped


def replay_through_oracle(calls, texts, targets, cues, N, g, K, budget, thr, seed):
    """``calls``: Recorder.calls of ONE solo search (grid batch, then that iteration's speculative verification batch
    if any, ...).  Returns (SearcherRef after search(), its keyframe timestamps)."""
    names = [t[0] for t in texts]
    it = iter(calls)
    pending = {}
    holder = {}

    def score_fn(kind, secs, rows, cols):
        if kind == "grid":
            c = next(it)
            assert (c["rows"], c["cols"]) == (rows, cols)
            nm = [[names[q] for q in range(len(names)) if (int(m) >> q) & 1] for m in c["mask"][0]]
            # the speculative verification batch of this iteration follows: every sampled frame whose cell lists a
            # target that is still remaining when the iteration STARTS (tstar_amd TStarSearcher._verify_launch)
            remaining = list(holder["ref"].remaining)
            cands = [i for i, x in enumerate(nm[:len(secs)]) if any(t in x for t in remaining)]
            pending.clear()
            if cands:
                v = next(it)
                assert v["conf"].shape[0] == len(cands)
                for j, i in enumerate(cands):
                    pending[secs[i]] = (v["conf"][j, 0], v["mask"][j, 0])
            return c["conf"][0].reshape(rows, cols), nm
        conf, m = pending[secs[0]]
        return np.array([[conf]]), [[names[q] for q in range(len(names)) if (int(m) >> q) & 1]]

    ref = S.SearcherRef(N, 1.0, list(targets), list(cues), score_fn, np.random.RandomState(seed), search_nframes=K,
                        image_grid_shape=(g, g), search_budget=budget, confidence_threshold=thr)
    holder["ref"] = ref
    ts_ref = ref.search()
    assert next(it, None) is None, "the HIP pipeline scored a batch the reference loop never asks for"
    return ref, ts_ref
