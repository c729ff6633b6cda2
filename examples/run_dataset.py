#!/usr/bin/env python
"""Dataset run in the shape of the reference's LVHaystackBench/run_TStar_onDataset.py:149-213, on MI355X:
items are sharded over the ranks (one process per GPU), each rank advances its items in lock-step
groups, the keyframe indices are all-gathered (RCCL) and rank 0 writes the reference's result JSON
(video_path, grounding_objects, keyframe_timestamps, keyframe_distribution).

Grounding (question -> target / cue objects) is a remote VLM call in the reference and is out of scope
here: every item carries its objects.  Videos are synthetic:// URLs (BASELINE configs[2]: the real
LV-Haystack split needs network).

  python examples/run_dataset.py --items 8 --out /tmp/results.json
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/run_dataset.py --items 32
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

QUESTIONS = [(["couch"], ["tv", "chair"]), (["dog"], ["leash", "park bench"]), (["red car"], ["road"]),
             (["laptop", "mug"], ["desk"])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=8)
    ap.add_argument("--nframes", type=int, default=600)
    ap.add_argument("--search-nframes", type=int, default=8)
    ap.add_argument("--grid", type=int, default=4)
    ap.add_argument("--lockstep", type=int, default=16)
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--out", default="./output/tstar_results.json")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tstar_amd.interface_heuristic import initialize_heuristic
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.lockstep import search_lockstep_groups
    from tstar_amd.results import make_result, save_results
    from tstar_amd.sharding import gather_keyframes, interleave_by_item, item_seed, shard_items

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        # torch.distributed is only the control plane (gloo); the one data collective -- the all-gather of keyframe rows -- goes over RCCL
        # through the library's own communicator (tstar_amd.sharding), or over gloo with TSTAR_BENCH_BACKEND=gloo / when RCCL cannot come up
        from tstar_amd import sharding as _shard
        _shard.PREFER_RCCL = os.environ.get("TSTAR_BENCH_BACKEND", "nccl") == "nccl"
        dist.init_process_group("gloo")

    items = [{"video_path": f"synthetic://n={args.nframes},seed={100 + i}", "targets": QUESTIONS[i % 4][0],
              "cues": QUESTIONS[i % 4][1]} for i in range(args.items)]
    heuristic = initialize_heuristic("owl-vit", synthetic_seed=0, max_batch=64, device=f"cuda:{local}")
    mine = shard_items(len(items), world, rank)
    rows, dists = [], {}
    # two lock-step groups alternate on the GPU: one group's host bookkeeping runs under the other's verification batch
    L = max(1, min(args.lockstep, 31))
    for g0 in range(0, len(mine), 2 * L):
        groups = [mine[g1:g1 + L] for g1 in range(g0, min(g0 + 2 * L, len(mine)), L)]
        sss = [[TStarSearcher(items[i]["video_path"], heuristic, list(items[i]["targets"]), list(items[i]["cues"]),
                              search_nframes=args.search_nframes, image_grid_shape=(args.grid, args.grid),
                              search_budget=1000, confidence_threshold=0.6,
                              rng=np.random.RandomState(item_seed(args.seed, i)), keep_visual_history=False) for i in group]
               for group in groups]
        for group, ss, res in zip(groups, sss, search_lockstep_groups(sss)):
            for i, s, (frames, ts) in zip(group, ss, res):
                rows.append([int(t) for t in ts])
                dists[i] = s.P_history[-1] if s.P_history else []
    gathered = gather_keyframes(rows, world, pad_to=(len(items) + world - 1) // world)
    if world > 1:
        gathered = interleave_by_item(gathered, len(items), world)
        parts = [None] * world                         # the per-second distributions (N floats per item) ride along
        dist.all_gather_object(parts, dists)
        dists = {k: v for d in parts for k, v in d.items()}
    if rank == 0:
        results = [make_result(it["video_path"], it["targets"], it["cues"], [float(t) for t in gathered[i]], dists.get(i, []))
                   for i, it in enumerate(items)]
        save_results(results, args.out)
        print(f"{len(results)} items -> {args.out}; item 0 keyframes {results[0]['keyframe_timestamps']}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
