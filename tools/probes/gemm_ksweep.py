#!/usr/bin/env python
"""Forward GEMM time vs K at fixed M, N: the intercept is the per-tile prologue/epilogue/tail cost, the slope the main-loop rate."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_gemm  # noqa: E402

M, N = int(sys.argv[1]), int(sys.argv[2])
for K in (64, 256, 1024, 2048, 4096, 8192):
    bench_gemm.run(M, N, K, iters=20)
