"""Encoder-shaped GEMMs (K = 768): persistent ping-pong kernel vs torch (hipBLASLt), interleaved.   python tools/probes/gemm_enc.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_gemm8p as b  # noqa: E402

for (M, N, K, act) in [(103777, 3072, 768, 2), (103777, 3072, 768, 0), (103777, 2304, 768, 0), (103777, 768, 3072, 0), (103777, 768, 768, 0),
                       (63040, 3072, 768, 3), (40960, 8192, 2048, 1), (40960, 8192, 2048, 0), (40960, 2048, 2048, 0)]:
    b.bench(M, N, K, act)
