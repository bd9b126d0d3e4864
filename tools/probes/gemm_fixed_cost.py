"""Per-tile fixed cost of the persistent GEMM: time ~ rounds * (a + b*K) over a K sweep at fixed M, N (20 full rounds of 256 tiles).
    MMGL_LIB_PATH=<ablated .so> python tools/probes/gemm_fixed_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402

M, N = 40960, 8192          # 160 x 32 tiles = 20 rounds of 256
res = []
for K in (256, 512, 768, 1024, 2048, 4096):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(x, w, out=y)
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.gemm_nt(x, w, out=y)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5 * 1e3)
    t = sorted(ts)[2]
    res.append((K, t))
    print(f"K={K:5d}  {t:8.1f} us  per tile-round {t / 20:6.2f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF", flush=True)
(k0, t0), (k1, t1) = res[2], res[-1]
b = (t1 - t0) / (k1 - k0) / 20
a = t0 / 20 - b * k0
print(f"{os.environ.get('MMGL_LIB_PATH', 'shipped build')}: fixed {a:.2f} us per tile + {b * 1e3:.2f} ns per k ({2 * 256 * 256 / b / 1e6 * 256:.0f} TF asymptotic)")
