#!/usr/bin/env python
"""Few-tile shapes (the reference's batch, M = 2560) under the current MMGL_GEMM_8P_SPLIT_MIN_UNITS / library.   one process per setting"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

SHAPES = [(2560, 2048, 2048), (2560, 2048, 8192), (2560, 2048, 6144), (2560, 4096, 4096), (1280, 2048, 2048), (5120, 2048, 2048), (2560, 6144, 2048)]
out = []
for M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm_nt(x, w, b, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ops.gemm_nt(x, w, b, out=y)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
    out.append(f"{M}x{N}x{K}: {best:6.1f} us")
print(f"min_units={os.environ.get('MMGL_GEMM_8P_SPLIT_MIN_UNITS', '24')} lib={os.path.basename(os.environ.get('MMGL_LIB_PATH', 'default'))} | " + " | ".join(out), flush=True)
