"""Attention kernel alone (vision shape: T = 577, 12 heads) at a few batch sizes; used by tools/pmc_attention_counters.sh.
    python tools/bench_attention.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tstar_amd import _lib

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
for B in [int(v) for v in sys.argv[1:]] or [16, 64, 256]:
    M = B * 577
    qkv = torch.randn(M, 2304, device="cuda")
    out = torch.empty(M, 768, device="cuda")
    f = lambda: _lib.check(lib.tstar_attention_f32(qkv.data_ptr(), out.data_ptr(), B, 577, 12, 0, None, s))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"attn B={B:3d} {ms:8.3f} ms  {4.0 * B * 12 * 577 * 577 * 64 / ms / 1e9:8.1f} TFLOP/s", flush=True)
