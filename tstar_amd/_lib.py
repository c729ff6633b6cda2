"""ctypes binding of libtstar_hip.so (include/tstar_hip.h).

There is NO CPU fallback: importing this module without the built library, or
calling a compute entry point without a HIP device, raises.  Build with
``python -m tstar_amd.build`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtstar_hip.so")


class TStarHipError(RuntimeError):
    pass


# name -> (restype, argtypes); must list EVERY symbol declared in include/tstar_hip.h
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_fp = C.POINTER(C.c_float)
SIGNATURES = {
    "tstar_last_error": (C.c_char_p, []),
    "tstar_abi_version": (_i, []),
    "tstar_owl_vision_blob_floats": (_sz, []),
    "tstar_owl_text_blob_floats": (_sz, []),
    "tstar_owl_create": (_i, [C.POINTER(_vp), _vp, _sz, _vp, _sz, _vp, _i, _i]),
    "tstar_owl_destroy": (_i, [_vp]),
    "tstar_owl_set_queries": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    "tstar_owl_set_queries_many": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tstar_owl_set_query_embeds": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    "tstar_owl_set_class_weights": (_i, [_vp, _i, _vp, _i, _vp]),
    "tstar_owl_get_query_embeds": (_i, [_vp, _i, _vp, _i, _vp]),
    "tstar_owl_score": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tstar_owl_score_lane": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tstar_owl_debug_preprocess": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "tstar_yolo_create": (_i, [C.POINTER(_vp), _vp, _sz, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i]),
    "tstar_yolo_destroy": (_i, [_vp]),
    "tstar_yolo_num_anchors": (_i, [_vp]),
    "tstar_yolo_set_text_feats": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "tstar_yolo_set_class_weights": (_i, [_vp, _i, _vp, _i, _vp]),
    "tstar_yolo_detect": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_float, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tstar_frames_to_grid": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "tstar_frames_resize": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "tstar_i420_to_nv12": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "tstar_nv12_to_rgb": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "tstar_searcher_create": (_i, [C.POINTER(_vp), _i, C.c_double, C.c_double]),
    "tstar_searcher_destroy": (_i, [_vp]),
    "tstar_searcher_apply_grid": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "tstar_searcher_set_spline": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tstar_searcher_sampler_prep": (_i, [_vp, _i, C.c_double, _vp, _vp]),
    "tstar_searcher_pop_prep": (_i, [_vp, _vp, _vp, _vp]),
    "tstar_searcher_window_spread": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tstar_searcher_visited": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "tstar_searcher_write": (_i, [_vp, _i, _vp, _vp]),
    "tstar_searcher_draw": (_i, [_vp, _vp, _i, _vp, _vp]),
    "tstar_searcher_exclude": (_i, [_vp, _vp, _i, _vp]),
    "tstar_searcher_set_scores": (_i, [_vp, _vp, _vp, _i, _vp]),
    "tstar_searcher_read_state": (_i, [_vp, _vp, _vp]),
    "tstar_searcher_read": (_i, [_vp, _i, _vp, _vp]),
    "tstar_comm_available": (_i, []),
    "tstar_comm_unique_id": (_i, [_vp]),
    "tstar_comm_create": (_i, [C.POINTER(_vp), _vp, _i, _i]),
    "tstar_comm_destroy": (_i, [_vp]),
    "tstar_allgather_i32": (_i, [_vp, _vp, _vp, _i, _vp]),
    "tstar_topk_seconds": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "tstar_ssim_pairwise": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "tstar_gemm_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tstar_gemm_f32_cfg": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tstar_gemm_bf16w": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tstar_gemm_bf16w2": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tstar_gemm_bf16w_pre": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tstar_gemm_f32x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tstar_pack_f32x3": (_i, [_vp, _vp, _i, _i, _vp]),
    "tstar_gemm_f32x3_pre": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tstar_layernorm_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tstar_draw_boxes": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "tstar_attention_split": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tstar_attention_x3": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tstar_attention_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "tstar_prof_enable": (_i, [_i]),
    "tstar_prof_read": (_i, [_i, _vp, _vp, _vp]),
    "tstar_prof_read_bytes": (_i, [_i, _vp]),
    "tstar_prof_read_totals": (_i, [_i, _vp, _vp]),
    "tstar_prof_mark": (_i, [_i, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the library (once) and bind every declared symbol; fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise TStarHipError(
            f"{LIB_PATH} is missing: build it with `python -m tstar_amd.build` "
            "(hipcc --offload-arch=gfx950). tstar_amd has no CPU fallback.")
    # torch ships its own libamdhip64: import it FIRST so this library binds to the same HIP
    # runtime (loading ours first pulls in /opt/rocm's copy and the two runtimes do not share
    # devices or memory).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().tstar_last_error()
        raise TStarHipError(f"{what or 'tstar_hip'} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
