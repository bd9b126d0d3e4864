#!/usr/bin/env python
"""The 128x128 kernel (csrc/gemm_mid.hip) against the other routes on the shapes of the reference's batch of 4 (M = 2560):
MMGL_GEMM_MID = 0 (off: 128x128 round-1 kernel / persistent kernel with K splits), 1 (default routing), 2 (every shape on it).
    python tools/probes/gemm_mid.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(2560, 2048, 2048, 0), (2560, 8192, 2048, 1), (2560, 2048, 8192, 0), (2560, 6144, 2048, 0), (2560, 2048, 6144, 0),
          (5120, 2048, 2048, 0), (10240, 2048, 2048, 0), (1280, 2048, 2048, 0), (6500, 768, 768, 0), (6500, 3072, 768, 2), (6500, 768, 3072, 0)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from mmgl_amd import ops
    for (M, N, K, act) in SHAPES:
        x = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm_nt(x, W, b, act=act, out=y)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(x, W, b, act=act, out=y)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print(f"   M={M:6d} N={N:5d} K={K:5d} act={act}: {best:7.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TF", flush=True)
else:
    for mode in os.environ.get("GEMM_MID_MODES", "0,1,2").split(","):
        print(f"MMGL_GEMM_MID={mode}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MMGL_GEMM_MID=mode), check=False)
