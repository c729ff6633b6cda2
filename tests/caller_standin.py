"""A stand-in for the reference's CALLERS of the hot path, written fresh for the tests (nothing here is reference
code): it exercises the two plug-in classes through exactly the call shapes the reference's drivers use, so a
mismatch in a keyword name, a return type or an attribute layout shows up as a test failure.

Call shapes reproduced (what is called, with which keywords, and what is read back):
  * /root/reference/TStar/TStarFramework.py:97-107   TStarSearcher(video_path=, target_objects=, cue_objects=,
        search_nframes=, image_grid_shape=(rows, cols), output_dir=, confidence_threshold=, search_budget=, heuristic=)
  * :116-124   all_frames, time_stamps = searcher.search(); len(all_frames)
  * :143-147   for idx, (frame, timestamp) in enumerate(zip(frames, timestamps)): f"{timestamp:.2f}"; frame HxWx3 u8 RGB
  * :152-157   image_grid_iters / detect_annotot_iters walked as [iteration][b] for b in range(len(image_grid_iters[0]))
  * :167       searcher.plot_score_distribution(save_path=...)
  * :171-187   initialize_heuristic(heuristic_type) shared by every item
  * /root/reference/LVHaystackBench/run_TStar_onDataset.py:128  time_stamps.sort()  (in place: a mutable list)
  * :139-144   result dict with video_searcher.P_history[-1]; :207-208 json.dump(results, indent=4, ensure_ascii=False)
The sampler draws from the process-global numpy generator, seeded once by the caller (val_qa_results.py:319 is the
only seed in the reference), and the visual history stays ON (the default), as under TStarFramework.
"""
import json
import os

import numpy as np


class StandinFramework:
    """Same constructor keywords and attribute names as TStarFramework (TStarFramework.py:26-52); no grounder:
    the question's objects are given."""

    def __init__(self, video_path, heuristic, target_objects, cue_objects, search_nframes=8, grid_rows=4, grid_cols=4,
                 output_dir="./output", confidence_threshold=0.6, search_budget=1000):
        self.video_path = video_path
        self.heuristic = heuristic
        self.target_objects, self.cue_objects = target_objects, cue_objects
        self.search_nframes = search_nframes
        self.grid_rows, self.grid_cols = grid_rows, grid_cols
        self.output_dir = output_dir
        self.confidence_threshold = confidence_threshold
        self.search_budget = search_budget
        os.makedirs(self.output_dir, exist_ok=True)
        self.saved = {"frames": [], "iterations": 0, "plot": None}

    def initialize_videoSearcher(self, searcher_cls):
        return searcher_cls(
            video_path=self.video_path,
            target_objects=self.target_objects,
            cue_objects=self.cue_objects,
            search_nframes=self.search_nframes,
            image_grid_shape=(self.grid_rows, self.grid_cols),
            output_dir=self.output_dir,
            confidence_threshold=self.confidence_threshold,
            search_budget=self.search_budget,
            heuristic=self.heuristic,
        )

    def perform_search(self, video_searcher, visualization=True):
        all_frames, time_stamps = video_searcher.search()
        if visualization:
            self._save_frames(all_frames, time_stamps)
            self._save_searching_iterations(video_searcher)
            plot_path = os.path.join(self.output_dir, "score_distribution.png")
            video_searcher.plot_score_distribution(save_path=plot_path)
            self.saved["plot"] = plot_path
        assert len(all_frames) == len(time_stamps)
        return all_frames, time_stamps

    def _save_frames(self, frames, timestamps):
        for idx, (frame, timestamp) in enumerate(zip(frames, timestamps)):
            name = f"frame_{idx}_at_{timestamp:.2f}s.jpg"                  # timestamps must format as floats
            assert frame.ndim == 3 and frame.shape[2] == 3 and frame.dtype == np.uint8
            bgr = np.ascontiguousarray(frame[:, :, ::-1])                  # what cv2.cvtColor(RGB2BGR) hands to imwrite
            self.saved["frames"].append((name, bgr.shape))

    def _save_searching_iterations(self, video_searcher):
        image_grid_iters = video_searcher.image_grid_iters
        detect_annotot_iters = video_searcher.detect_annotot_iters
        for b in range(len(image_grid_iters[0])):
            images = [image_grid_iter[b] for image_grid_iter in image_grid_iters]
            anno_images = [detect_annotot_iter[b] for detect_annotot_iter in detect_annotot_iters]
            assert len(images) == len(anno_images)
            for im, an in zip(images, anno_images):                        # what save_as_gif consumes: HxWx3 uint8
                assert im.dtype == np.uint8 and im.ndim == 3 and an.shape == im.shape and an.dtype == np.uint8
            self.saved["iterations"] = len(anno_images)


def run_item(searcher_cls, heuristic, data_item, args):
    """One dataset item the way get_TStar_search_results does it (run_TStar_onDataset.py:108-146)."""
    fw = StandinFramework(video_path=data_item["video_path"], heuristic=heuristic, target_objects=data_item["targets"],
                          cue_objects=data_item["cues"], search_nframes=args["search_nframes"], grid_rows=args["grid_rows"],
                          grid_cols=args["grid_cols"], output_dir=args["output_dir"],
                          confidence_threshold=args["confidence_threshold"], search_budget=args["search_budget"])
    video_searcher = fw.initialize_videoSearcher(searcher_cls)
    all_frames, time_stamps = fw.perform_search(video_searcher, visualization=True)
    time_stamps.sort()                                                     # in place: needs a mutable list
    result = {
        "video_path": data_item["video_path"],
        "grounding_objects": {"target_objects": data_item["targets"], "cue_objects": data_item["cues"]},
        "keyframe_timestamps": time_stamps,
        "keyframe_distribution": video_searcher.P_history[-1],
    }
    return result, fw, video_searcher, all_frames


def dump_results(results, path):
    with open(path, "w", encoding="utf-8") as f_out:
        json.dump(results, f_out, indent=4, ensure_ascii=False)
