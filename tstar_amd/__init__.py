"""tstar_amd — MI355X-native T* keyframe-search hot path.

Host-side mirror of the reference plug-in surface
(/root/reference/TStar/interface_heuristic.py, interface_searcher.py) over a
C-ABI library of hand-written gfx950 HIP kernels (include/tstar_hip.h).
"""
__version__ = "0.1.0"
