"""Visual-history outputs of a search: what the reference's framework writes next to a result.

Host-side and cosmetic -- nothing here is on the scoring path.  Mirrors, with the same file names and Pillow calls:
  * save_as_gif             /root/reference/TStar/utilites.py:84-102 (1 frame per second, endless loop)
  * encode_image_to_base64  utilites.py:15-37 (JPEG, Pillow defaults)
  * save_frames             TStarFramework._save_frames, TStarFramework.py:136-147 (frames/frame_{i}_at_{t:.2f}s.jpg; the
                            reference writes them with cv2.imwrite, this writes the same RGB content with Pillow at
                            OpenCV's default JPEG quality 95 -- the encoders differ, so the files are not byte-equal)
  * save_searching_iterations  TStarFramework._save_searching_iterations, :149-162 (search_iterations.gif from
                            searcher.detect_annotot_iters[i][b]; like the reference every batch element b overwrites
                            the same file, so the GIF shows the last one)
  * plot_and_save_scores    TStarFramework._plot_and_save_scores, :164-170
"""
from __future__ import annotations

import base64
import io
import os
from typing import List, Sequence

import numpy as np
from PIL import Image


def encode_image_to_base64(image) -> str:
    try:
        if isinstance(image, np.ndarray):
            image = Image.fromarray(image)
        if not isinstance(image, Image.Image):
            raise ValueError("Input must be a PIL.Image or numpy.ndarray")
        buf = io.BytesIO()
        image.save(buf, format="JPEG")
        return base64.b64encode(buf.getvalue()).decode("utf-8")
    except Exception as e:                                   # the reference wraps every failure the same way
        raise ValueError(f"Error encoding image: {str(e)}")


def save_as_gif(images: Sequence[np.ndarray], output_gif_path: str) -> None:
    frames = [Image.fromarray(np.asarray(im).astype("uint8")) for im in images]
    frames[0].save(output_gif_path, save_all=True, append_images=frames[1:], duration=1000, loop=0)
    print(f"Saved GIF: {output_gif_path}")


def save_frames(frames: Sequence[np.ndarray], timestamps: Sequence[float], output_dir: str) -> List[str]:
    frame_dir = os.path.join(output_dir, "frames")
    os.makedirs(frame_dir, exist_ok=True)
    paths = []
    for idx, (frame, ts) in enumerate(zip(frames, timestamps)):
        path = os.path.join(frame_dir, f"frame_{idx}_at_{ts:.2f}s.jpg")
        Image.fromarray(np.asarray(frame).astype("uint8")).save(path, format="JPEG", quality=95)
        paths.append(path)
    return paths


def save_searching_iterations(searcher, output_dir: str) -> str:
    grids, annos = searcher.image_grid_iters, searcher.detect_annotot_iters
    path = os.path.join(output_dir, "search_iterations.gif")
    for b in range(len(grids[0])):
        save_as_gif(images=[it[b] for it in annos], output_gif_path=path)
    return path


def plot_and_save_scores(searcher, output_dir: str) -> str:
    path = os.path.join(output_dir, "score_distribution.png")
    searcher.plot_score_distribution(save_path=path)
    return path
