#!/usr/bin/env python
"""Outputs of one to three rounds of 256x256 tiles plus a small remainder (the reference's batch: 2560 x 8192 = 320 tiles on 256 CUs):
whole tiles + K-split remainder (default) against MMGL_GEMM_8P_HYBRID=0, one process per setting.   python tools/probes/gemm_hybrid.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

SHAPES = [(2560, 8192, 2048, 1, False), (2560, 8192, 2048, 0, True), (2560, 2048, 8192, 0, False), (2560, 6144, 2048, 0, False),
          (5120, 8192, 2048, 1, False), (5120, 6144, 2048, 0, False), (1280, 8192, 2048, 1, False)]
out = []
for M, N, K, act, zm in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    z = torch.randn(M, N, device="cuda").bfloat16() if zm else None
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm_nt(x, w, b, None, z, act=act, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.gemm_nt(x, w, b, None, z, act=act, out=y)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    out.append(f"{M}x{N}x{K}{'+z' if zm else ''}: {best:6.1f} us")
print(f"hybrid={os.environ.get('MMGL_GEMM_8P_HYBRID', '3')} | " + " | ".join(out), flush=True)
