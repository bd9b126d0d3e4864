#!/usr/bin/env python
"""zmask / residual epilogue shapes of the step under the current library (MMGL_LIB_PATH).   one process per setting"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

out = []
for M, N, K, zm, rs in [(40960, 8192, 2048, True, False), (40960, 2048, 2048, False, True), (40960, 2048, 8192, False, True), (40960, 8192, 2048, False, False)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    z = torch.randn(M, N, device="cuda").bfloat16() if zm else None
    r = torch.randn(M, N, device="cuda").bfloat16() if rs else None
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm_nt(x, w, None, r, z, out=y)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_nt(x, w, None, r, z, out=y)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append(f"{M}x{N}x{K}{'+z' if zm else ''}{'+r' if rs else ''}: {best:7.1f} us")
print(f"lib={os.path.basename(os.environ.get('MMGL_LIB_PATH', 'default'))} | " + " | ".join(out), flush=True)
