"""L3 free-running agreement: the HIP pipeline closed-loop against the CPU oracle pipeline closed-loop.

Both run the whole search on their own scores (no teacher forcing) from the same sampler seed; because the sampler
thresholds at a percentile and draws from the score-derived distribution, a last-place difference in one confidence can
change a later draw (the loop is chaotic), so equality is REPORTED, not gated: per seed the number of iterations whose
sampled seconds coincide, the first diverging iteration, the largest confidence difference over the coinciding prefix and
whether the keyframes are equal.  The oracle pipeline itself follows the reference end to end (goldens G9 / G9b,
tests/test_oracle_searcher.py::test_oracle_pipeline_free_running_vs_reference_end_to_end).

    python tests/l3_agreement_report.py [--seeds 8] [--modes f32,f32x3] > gpurun_out/l3_agreement.md      (GPU box; ~15 s of CPU per seed)

Lives under tests/ because it drives the CPU oracle (test infrastructure); tests/test_gpu_searcher.py runs a two-seed pass of it.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--modes", default="f32,f32x3")
    ap.add_argument("--nframes", type=int, default=3600)
    ap.add_argument("--grid", type=int, default=4)
    ap.add_argument("--budget", type=float, default=0.035, help="fraction of the frames: 0.035 * 3600 = 126 -> 8 iterations of 16")
    run(ap.parse_args())


def run(args):
    """Prints the report; returns {mode: [iterations on the oracle's trajectory, oracle iterations, searches with equal keyframes]}."""
    import torch
    torch.set_num_threads(16)
    from oracle import cpu_pipeline, searcher_ref as S
    from tstar_amd import weights as W
    from tstar_amd.interface_heuristic import OWLInterface
    from tstar_amd.interface_searcher import TStarSearcher
    from tstar_amd.tokenizer import encode_queries
    from tstar_amd.video import synthetic_frames_numpy, synthetic_video
    N, g, K = args.nframes, args.grid, 8
    targets, cues = ["couch"], ["tv", "chair"]
    modes = args.modes.split(",")
    hs = {m: OWLInterface(synthetic_seed=0, max_batch=32, weights_dtype=m) for m in modes}
    det = cpu_pipeline.CpuOwlDetector(W.synthetic_state_dict(0), faithful=False)
    texts = [[t] for t in targets + cues] + [[" "]]
    ids, am = encode_queries(texts)
    det.reparameterize_object_list(targets, cues, ids, am)
    print(f"# L3 free-running agreement: HIP pipeline vs CPU oracle pipeline, both closed-loop (N = {N}, grid {g}x{g}, K = {K}, "
          f"budget {args.budget} = {int(min(1000, N * args.budget))} frames, threshold 0.6)\n")
    print("| seed (video, sampler) | mode | iterations HIP / oracle | iterations on the same trajectory | first divergence | max \\|conf - oracle conf\\| on the common prefix | keyframes equal |")
    print("|---|---|---:|---:|---|---:|---|")
    tot = {m: [0, 0, 0] for m in modes}
    for sd in range(args.seeds):
        vseed, sseed = 300 + sd, 7000 + 13 * sd
        t0 = time.time()
        olog = []
        fn = cpu_pipeline.make_score_fn(det, lambda secs: synthetic_frames_numpy(list(secs), N, seed=vseed), {"couch": 1.0, "tv": 0.5, "chair": 0.5}, olog)
        ref = S.SearcherRef(N, 1.0, targets, cues, fn, np.random.RandomState(sseed), search_nframes=K, image_grid_shape=(g, g),
                            search_budget=args.budget, confidence_threshold=0.6)
        ots = ref.search()
        osecs = [it["secs"] for it in ref.trace]
        oconf = [c["conf"] for c in olog if c["kind"] == "grid"]
        store = synthetic_video(N, seed=vseed)
        for m in modes:
            h = hs[m]
            s = TStarSearcher(store, h, targets, cues, search_nframes=K, image_grid_shape=(g, g), search_budget=args.budget,
                              confidence_threshold=0.6, rng=np.random.RandomState(sseed), keep_visual_history=False)
            log, confs = [], []
            orig, osb = s.sample_frames, h.score_batch
            s.sample_frames = lambda num, _o=orig: (lambda r: (log.append(list(r[0])), r)[1])(_o(num))

            def rec(d_images, rows, cols, image_sets=None, _o=osb, **kw):
                r = _o(d_images, rows, cols, image_sets=image_sets, **kw)
                if rows == g:
                    confs.append(r.cell_conf.cpu().numpy()[0].reshape(g, g))
                return r
            h.score_batch = rec
            try:
                _, ts = s.search()
            finally:
                h.score_batch = osb
            same, worst = 0, 0.0
            for it in range(min(len(log), len(osecs))):
                if log[it] != osecs[it]:
                    break
                worst = max(worst, float(np.abs(confs[it] - oconf[it]).max()))
                same += 1
            full = same == len(osecs) == len(log)
            eq = [float(t) for t in ts] == ots
            tot[m][0] += same; tot[m][1] += len(osecs); tot[m][2] += int(eq)
            print(f"| {vseed}, {sseed} | {m} | {len(log)} / {len(osecs)} | {same} | {'none' if full else 'iteration ' + str(same)} | {worst:.2e} | {'yes' if eq else 'no'} |")
        print(f"<!-- seed {sd}: {time.time() - t0:.0f} s -->", file=sys.stderr)
    print()
    for m in modes:
        print(f"* **{m}**: {tot[m][0]} of {tot[m][1]} iterations on the oracle's trajectory ({100.0 * tot[m][0] / max(tot[m][1], 1):.1f} %), "
              f"keyframes equal in {tot[m][2]} of {args.seeds} searches.")
    return tot


if __name__ == "__main__":
    main()
