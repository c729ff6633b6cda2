#!/usr/bin/env python
"""bench.py -- T* keyframe-search hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete keyframe search (TStarSearcher.search(): iterative sampling, grid
scoring, verification, distribution updates, final K=8 keyframes) over one 3600-frame
synthetic video that is already resident in HBM -- BASELINE.json configs[1]: "Same single
video on 1xMI355X, HIP OWL-ViT-B/32 scorer, batch=256 frames/iter" (grid 16x16).  With N > 1
(torch.distributed, one rank per GPU, RCCL) every rank searches its own stream of independent
(video, question) items (weak scaling, no data-path collective) and the final keyframe indices
are all-gathered once inside the timed region.  Within a rank, items advance in
lock-step groups of --lockstep (default 4): iteration t of every item of the group shares one
detector batch, each image scored against its own question (tstar_amd/lockstep.py); results are
bit-identical to one-by-one searches.

Prints ONE JSON line on rank 0.  value = candidate frames scored / s over all ranks; a frame
is scored each time it contributes a confidence to the searcher: g*g per grid call + 1 per
verification call.  Weights are seeded synthetic OWL-ViT-B/32 (no checkpoint
offline), data synthetic.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROF_STRIDE = 5
BF16_MFMA_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA peak (same guide)
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
N_FRAMES, FRAME_H, FRAME_W = 3600, 360, 640
TARGETS, CUES = ["couch"], ["tv", "chair"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=16, help="grid side g (g*g frames per iteration)")
    ap.add_argument("--max-batch", type=int, default=256, help="detector images per forward chunk")
    ap.add_argument("--nframes", type=int, default=N_FRAMES)
    ap.add_argument("--search-nframes", type=int, default=8)
    ap.add_argument("--weights", choices=["f32", "bf16", "f32_split"], default="f32",
                    help="bf16 = BASELINE config 5 (bf16-rounded weights, exact-split bf16 MFMA GEMMs); with "
                         "--nframes 14400 --grid 15 --search-nframes 32 this is configs[4].  f32_split = fp32 checkpoint "
                         "with every operand carried as two bf16 terms on the bf16 matrix pipe (opt-in)")
    ap.add_argument("--concurrency", type=int, default=1,
                    help="independent searches in flight per GPU (host threads, one HIP stream + one scorer "
                         "workspace each); 2 fills kernel tails and gives ~+5 %% throughput, but overlapping "
                         "launches inflate per-launch durations, so the roofline leg is reported at 1")
    ap.add_argument("--lockstep", type=int, default=4,
                    help="independent (video, question) items advanced in lock-step per detector batch "
                         "(tstar_amd.lockstep; results identical to one-by-one searches); 1 = one at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget for the CPU baseline sample")
    return ap.parse_args()


def make_searcher(heuristic, store, g, seed, k=8):
    from tstar_amd.interface_searcher import TStarSearcher
    return TStarSearcher(store, heuristic, list(TARGETS), list(CUES), search_nframes=k, image_grid_shape=(g, g),
                         search_budget=1000, confidence_threshold=0.6, rng=np.random.RandomState(seed),
                         keep_visual_history=False)


def run_group(heuristic, store, g, seeds, k=8):
    """One lock-step group of independent searches (a single search when len(seeds) == 1)."""
    from tstar_amd.lockstep import search_lockstep
    ss = [make_searcher(heuristic, store, g, sd, k) for sd in seeds]
    if len(ss) == 1:
        return [(ss[0], ss[0].search()[1])]
    res = search_lockstep(ss)
    return [(s_, r[1]) for s_, r in zip(ss, res)]


def cpu_baseline(args, stats):
    """Reference-faithful CPU path (oracle = "port") on the host cores: one grid call and a few
    verification calls are timed, then extrapolated to the GPU run's call mix."""
    import torch
    from oracle import cpu_pipeline, owl_ref, resize_ref, searcher_ref
    from tstar_amd import weights as W
    from tstar_amd.tokenizer import encode_queries
    from tstar_amd.video import synthetic_frames_numpy
    g = args.grid
    det = cpu_pipeline.CpuOwlDetector(W.synthetic_state_dict(0), faithful=True)
    texts = [[o] for o in TARGETS + CUES] + [[" "]]
    ids, am = encode_queries(texts)
    det.reparameterize_object_list(TARGETS, CUES, ids, am)
    o2w = {**{t: 1.0 for t in TARGETS}, **{c: 0.5 for c in CUES}}
    n = g * g
    secs = list(np.arange(0, args.nframes, args.nframes // n)[:n])
    frames = synthetic_frames_numpy(secs, args.nframes, FRAME_H, FRAME_W, seed=0)    # "decoded" frames, untimed
    # The frame -> grid / verification-size resizes are cv2.resize calls in the reference (~1 ms per frame);
    # cv2 is not in this image and the numpy restatement of it takes ~50 ms per frame, so those images are
    # prepared UNTIMED (this favours the CPU).  Timed per detector call: Pillow bicubic to 768x768 (the real
    # library when importable), rescale/normalise, [text tower,] vision tower + heads, post-process, grid-cell
    # aggregation.
    grid_img = resize_ref.frames_to_grid(list(frames), g, g)
    nver = min(n, 64)
    ver_imgs = [resize_ref.cv_bilinear_resize(frames[i], 600, 285) for i in range(nver)]
    names = det.texts

    def call(d, img, rows, cols):
        xyxy, lab, conf, _ = d.inference_detector(img, library_speed=True)
        return searcher_ref.image_grid_score(xyxy, lab, conf, names, o2w, img.shape[0], img.shape[1], rows, cols)

    call(det, ver_imgs[0], 1, 1)                     # untimed warm-up (thread pools, first-touch allocations)
    # thread count: torch's default (all hardware threads) is far from the fastest on a many-core host for
    # batch-1 forwards; probe a few counts on one verification call each and keep the best for BOTH baselines
    nt_all = torch.get_num_threads()
    best_nt, best_t = nt_all, None
    for nt in sorted({min(c, nt_all) for c in (8, 16, 32, 64, nt_all)}):
        torch.set_num_threads(nt)
        call(det, ver_imgs[0], 1, 1)
        t0 = time.perf_counter()
        call(det, ver_imgs[1 % nver], 1, 1)
        t_ = time.perf_counter() - t0
        if best_t is None or t_ < best_t:
            best_nt, best_t = nt, t_
    torch.set_num_threads(best_nt)
    t0 = time.perf_counter()
    call(det, grid_img, g, g)
    t_grid = time.perf_counter() - t0
    nv, t_ver = 0, 0.0
    deadline = time.perf_counter() + max(1.0, args.cpu_seconds - t_grid)
    while nv < nver and time.perf_counter() < deadline:
        t0 = time.perf_counter()
        call(det, ver_imgs[nv], 1, 1)
        t_ver += time.perf_counter() - t0
        nv += 1
    t_ver /= max(nv, 1)
    per_video = stats["grid_calls"] * t_grid + stats["verify_calls"] * t_ver
    frames_scored = stats["grid_calls"] * n + stats["verify_calls"]
    # CPU(O) of SURVEY 8d: the same restatement with this build's own call structure (text tower cached,
    # verification frames batched) -- what the CPU does when only the arithmetic, not the reference's call
    # pattern, is kept
    det_o = cpu_pipeline.CpuOwlDetector(W.synthetic_state_dict(0), faithful=False)
    det_o.reparameterize_object_list(TARGETS, CUES, ids, am)
    det_o.query_embeds()
    t0 = time.perf_counter()
    call(det_o, grid_img, g, g)
    t_grid_o = time.perf_counter() - t0
    t_ver_o, vb = None, 1
    for b_ in (1, 8):                                # best of: one frame per forward, eight per forward
        b_ = min(b_, nver)
        reps = 3 if b_ == 1 else 1
        t0 = time.perf_counter()
        for r_ in range(reps):
            px = np.stack([det_o.preprocess(ver_imgs[(r_ + i) % nver], library_speed=True) for i in range(b_)])
            owl_ref.detect(px, det_o.query_embeds(), det_o.wv, 285, 600, query_mask=det_o.ids[:, 0] > 0)
        t_b = (time.perf_counter() - t0) / (b_ * reps)
        if t_ver_o is None or t_b < t_ver_o:
            t_ver_o, vb = t_b, b_
    per_video_o = stats["grid_calls"] * t_grid_o + stats["verify_calls"] * t_ver_o
    return {
        "value": frames_scored / per_video, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"1 grid call ({n} frames, {t_grid:.3f} s) + {nv} verification calls ({t_ver:.3f} s each) of the same "
                  f"workload, reference-faithful (text tower per call, batch-1 verification), extrapolated to the GPU "
                  f"run's call mix ({stats['grid_calls']} grid + {stats['verify_calls']} verification calls per video); "
                  f"{best_nt} torch threads (fastest of 8/16/32/64/{nt_all} on this host); "
                  f"the cv2.resize steps (not importable here) are prepared untimed",
        "sec_per_video": per_video,
        "own_batching": {"value": frames_scored / per_video_o, "unit": "frames/s", "sec_per_video": per_video_o,
                         "sample": f"1 grid call ({t_grid_o:.3f} s, cached query embeddings) + verification forwards of {vb} "
                                   f"frame(s) ({t_ver_o:.3f} s per frame; best of batch 1 and 8), same extrapolation"},
    }


def main():
    args = parse()
    # the searcher prints progress like the reference ("Found target ...", sampler warnings): keep stdout
    # clean for the ONE JSON line
    real_stdout = sys.stdout
    sys.stdout = sys.stderr
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # one rank per GPU over RCCL; TSTAR_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a
    # single-GPU box (ranks share the device, collectives go through host tensors)
    backend = os.environ.get("TSTAR_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    cdev = "cuda" if backend == "nccl" else "cpu"

    from tstar_amd import _lib
    from tstar_amd.interface_heuristic import OWLInterface
    from tstar_amd.sharding import gather_keyframes
    from tstar_amd.video import synthetic_video
    lib = _lib.load()

    import queue
    import threading
    conc = max(1, min(args.concurrency, args.steps))
    heuristics = [OWLInterface(synthetic_seed=0, max_batch=args.max_batch, device=f"cuda:{local_rank}",
                               weights_dtype=args.weights) for _ in range(conc)]
    streams = [torch.cuda.Stream() for _ in range(conc)]
    store = synthetic_video(args.nframes, FRAME_H, FRAME_W, seed=0)
    g = args.grid

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_many(seeds):
        """Run one search per seed, `conc` at a time; returns per-search (searcher, timestamps, seconds)."""
        q = queue.Queue()
        L = max(1, min(args.lockstep, 31))
        for i in range(0, len(seeds), L):
            q.put((i, seeds[i:i + L]))
        out = [None] * len(seeds)
        errs = []

        def worker(w):
            torch.cuda.set_device(local_rank)
            try:
                with torch.cuda.stream(streams[w]):
                    while True:
                        try:
                            i, sd = q.get_nowait()
                        except queue.Empty:
                            break
                        t1 = time.perf_counter()
                        grp = run_group(heuristics[w], store, g, sd, args.search_nframes)
                        streams[w].synchronize()
                        for j, (s_, ts_) in enumerate(grp):
                            out[i + j] = (s_, ts_, time.perf_counter() - t1)
            except Exception as e:                      # surface worker failures in the main thread
                errs.append(e)

        if conc == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(w,)) for w in range(conc)]
            [t_.start() for t_ in th]
            [t_.join() for t_ in th]
        if errs:
            raise errs[0]
        return out

    # untimed warm-up: at least one FULL lock-step group, so the timed region meets no first-use cost
    # (kernel attributes, resample tables, workspace growth) at its own batch sizes
    n_warm = 0 if args.warmup <= 0 else max(args.warmup, min(args.lockstep, 31) * conc)
    run_many([10_000 + rank * 1000 + w for w in range(n_warm)])
    # latency of ONE search running alone (the "sec/video to 8 keyframes" half of the metric, untimed):
    solo_latency = None
    if n_warm > 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_group(heuristics[0], store, g, [20_000 + rank], args.search_nframes)
        torch.cuda.synchronize()
        solo_latency = time.perf_counter() - t1
    barrier()
    # time every 5th GEMM / attention launch with HIP event pairs (5 is co-prime with the 4-GEMM layer
    # pattern and the 52-GEMM forward, so every shape is sampled evenly); timing all of them costs 2.2 %
    _lib.check(lib.tstar_prof_enable(0 if os.environ.get('TSTAR_BENCH_NO_PROF') else PROF_STRIDE))
    t0 = time.perf_counter()
    res = run_many([2025 + rank * args.steps + k for k in range(args.steps)])
    frames = sum(r[0].frames_scored for r in res)
    grid_calls = sum(r[0].iterations for r in res)
    verify_calls = sum(r[0].detector_calls - r[0].iterations for r in res)
    images = sum(r[0].device_images_scored for r in res)
    keys = [[int(t) for t in r[1]] for r in res]
    latency = sum(r[2] for r in res) / len(res)
    all_keys = gather_keyframes(keys, world)            # RCCL all-gather of the keyframe indices (N > 1)
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(frames), float(images)], dtype=torch.float64, device=cdev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, frames_all, images_all = tmax[0].item(), t[1].item(), t[2].item()
    else:
        frames_all, images_all = float(frames), float(images)

    # roofline of the dominant kernel (gemm_f32_kernel): HIP events on the launch stream, this rank
    n_l, ms, fl = C.c_longlong(0), C.c_double(0), C.c_double(0)
    _lib.check(lib.tstar_prof_read(0, C.byref(n_l), C.byref(ms), C.byref(fl)))
    a_l, a_ms, a_fl = C.c_longlong(0), C.c_double(0), C.c_double(0)
    _lib.check(lib.tstar_prof_read(1, C.byref(a_l), C.byref(a_ms), C.byref(a_fl)))
    _lib.check(lib.tstar_prof_enable(0))
    achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    # HBM-side traffic of the same kernel: rocprofv3 PMC passes of this command cannot run inside the
    # timed process, so the per-launch figure measured with `tools/rocpd_traffic.py` is read from the
    # committed summary (profiles/); None if it has not been collected.
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "r01_pmc_gemm_traffic.json")
    if os.path.isfile(tp):
        try:
            tj = json.load(open(tp))
            traffic, traffic_src = tj["bytes_per_launch_corrected"], "profiles/r01_pmc_gemm_traffic.json"
        except Exception:
            pass

    # f32 weights: native f32 MFMA, algorithmic = executed flops.  bf16 weights: each algorithmic product is
    # three bf16 MFMA products (exact activation split), priced against the dense bf16 peak.
    if args.weights in ("bf16", "f32_split"):
        gemm_kernel, peak, exec_mult = f"gemm_f32_kernel<WMODE={1 if args.weights == 'bf16' else 2}> (3 x v_mfma_f32_32x32x16_bf16 per K=16)", BF16_MFMA_PEAK_TFLOPS, 3.0
    else:
        gemm_kernel, peak, exec_mult = "gemm_f32_kernel (v_mfma_f32_32x32x2_f32)", FP32_MFMA_PEAK_TFLOPS, 1.0
    if rank == 0:
        out = {
            "metric": "candidate frames scored/sec (whole node) + sec/video to 8 keyframes, 1h@1fps",
            "value": frames_all / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "f32 activations x bf16 weights (exact 3-term split on the bf16 MFMA pipe)",
                      "f32_split": "f32 operands as 2 bf16 terms each (16 significand bits), 3 bf16 MFMA products, f32 accumulate"}[args.weights],
            "data": "synthetic",
            "config": {
                "collective_backend": (backend if world > 1 else None),
                "workload": f"{'configs[1]' if args.weights == 'f32' and args.nframes == N_FRAMES else 'variant'}: {args.nframes}-frame {FRAME_H}x{FRAME_W} synthetic RGB video resident in HBM, 1 question "
                            f"(targets {TARGETS}, cues {CUES}), OWL-ViT-B/32 {args.weights} weights (seeded synthetic), grid {g}x{g} = "
                            f"{g * g} frames/iter, search_nframes={args.search_nframes}, threshold 0.6, budget 1000",
                "sec_per_video": dt / args.steps, "videos_per_rank": args.steps, "searches_in_flight_per_gpu": conc, "lockstep_items_per_batch": max(1, min(args.lockstep, 31)),
                "mean_search_latency_sec": latency, "single_search_alone_latency_sec": solo_latency,
                "grid_calls_per_video": grid_calls / args.steps, "verify_calls_per_video": verify_calls / args.steps,
                "detector_images_per_video": images / args.steps, "max_batch": args.max_batch,
                "keyframes_rank0_step0": keys[0], "gathered_keyframe_rows": len(all_keys),
            },
            "roofline": {
                "kernel": gemm_kernel, "bound": "mfma", "achieved": achieved * exec_mult,
                "peak": peak, "unit": "TFLOP/s", "frac": achieved * exec_mult / peak, "traffic": traffic,
                "achieved_algorithmic": achieved,
                "traffic_unit": "bytes per launch (L2 fabric side: FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)",
                "traffic_source": traffic_src,
                "launches_timed": n_l.value, "timed_every_nth_launch": PROF_STRIDE, "avg_launch_ms": ms.value / max(n_l.value, 1),
                "avg_launch_gflop": fl.value / max(n_l.value, 1) / 1e9,
                "time_share_of_step": ms.value * PROF_STRIDE * 1e-3 / dt / max(conc, 1),
                "attention_f32_kernel": {"achieved": (a_fl.value / (a_ms.value * 1e-3) / 1e12) if a_ms.value > 0 else 0.0,
                                         "launches_timed": a_l.value, "time_share_of_step": a_ms.value * PROF_STRIDE * 1e-3 / dt / max(conc, 1)},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, {"grid_calls": grid_calls / args.steps,
                                                      "verify_calls": verify_calls / args.steps})
            out["config"]["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
