"""Small-M GEMMs (the reference's batch 4: M = 2560) through mmgl_gemm_nt: which kernel wins below / above the tile threshold.
Run under MMGL_GEMM_8P_MIN_TILES=<n> to force the ping-pong kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = ["x"]
import tools.bench_gemm8p as b  # noqa: E402

for (M, N, K, act) in [(2560, 2048, 2048, 0), (2560, 6144, 2048, 0), (2560, 8192, 2048, 1), (2560, 2048, 8192, 0), (2560, 50272, 2048, 0),
                       (5120, 2048, 2048, 0), (10240, 2048, 2048, 0), (20480, 2048, 2048, 0)]:
    b.bench(M, N, K, act, rounds=3, iters=20)
