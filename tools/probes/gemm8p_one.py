"""A few launches of the ping-pong GEMM at one frozen-path shape (target of the PMC scripts).  python tools/probes/gemm8p_one.py [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (40960, 6144, 2048)
x = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm_nt(x, W, b, out=y)
src = torch.empty(128 << 20, dtype=torch.bfloat16, device="cuda").normal_()        # 256 MiB calibration copy
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
print(f"gemm M={M} N={N} K={K}: algorithmic bytes {2 * (M * K + N * K + M * N)} per launch; calibration copy 268435456 B read + written")
