"""Few-tile GEMMs of the reference's batch-4 step: persistent kernel with K splits vs the 128x128 kernel vs torch.
    python tools/probes/gemm_small.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_gemm8p as b  # noqa: E402

for (M, N, K, act) in [(2560, 2048, 2048, 0), (2560, 2048, 8192, 0), (2560, 2048, 6144, 0), (2560, 8192, 2048, 1), (2560, 6144, 2048, 0),
                       (6500, 768, 3072, 0), (6500, 768, 768, 0), (6500, 3072, 768, 2), (3940, 768, 3072, 0)]:
    b.bench(M, N, K, act)
