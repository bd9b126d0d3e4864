#!/usr/bin/env python
"""Time the default step's seven GEMM shapes through mmgl_gemm_nt under the current environment / MMGL_LIB_PATH (one process per setting:
the switches are read once).   python tools/probes/gemm_step_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

SHAPES = [(40960, 2048, 2048, 0, False), (40960, 6144, 2048, 0, False), (40960, 8192, 2048, 1, False),
          (40960, 2048, 8192, 0, True), (43520, 2048, 2048, 0, False), (16384, 2048, 2048, 0, False), (40960, 2048, 768, 0, False)]


def main():
    tag = os.environ.get("MMGL_LIB_PATH", "default library")
    tot = 0.0
    out = []
    for M, N, K, act, resid in SHAPES:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        r = torch.randn(M, N, device="cuda").bfloat16() if resid else None
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm_nt(x, w, b, r, None, act=act, out=y)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(x, w, b, r, None, act=act, out=y)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        tot += best
        out.append(f"{M}x{N}x{K}: {best:7.1f} us {2.0 * M * N * K / best / 1e6:6.0f} TF")
    print(tag, "| total", f"{tot:8.1f} us |", " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
