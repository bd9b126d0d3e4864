"""Interleaved A/B of the output row pitch for the fc1-shaped GEMM (N = 8192): pitch 8192 vs 8320 vs 9216 elements, rounds
alternating in one process.   python tools/probes/gemm_pitch2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402

M, N, K = 40960, 8192, 2048
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
bufs = {p: torch.empty(M, p, device="cuda", dtype=torch.bfloat16) for p in (8192, 8320, 9216)}
res = {p: [] for p in bufs}
for p, buf in bufs.items():
    for _ in range(3):
        ops.gemm_nt(x, w, b, act=1, out=buf[:, :N])
for _ in range(6):
    for p, buf in bufs.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.gemm_nt(x, w, b, act=1, out=buf[:, :N])
        e.record()
        torch.cuda.synchronize()
        res[p].append(s.elapsed_time(e) / 5 * 1e3)
for p, v in res.items():
    v = sorted(v)
    print(f"pitch {p}: median {v[len(v) // 2]:.1f} us  min {v[0]:.1f}  max {v[-1]:.1f}")
