#!/usr/bin/env python
"""1.25-round outputs at the reference's batch (M = 2560, N = 8192: 320 tiles of 256 x 256 on 256 CUs): the whole GEMM in one call
against the same rows as two calls -- 2048 rows (one exact round of the persistent kernel) + 512 rows (few-tile kernel).
    python tools/probes/gemm_msplit.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K, m1 in [(2560, 8192, 2048, 2048), (2560, 2048, 8192, 2048), (2560, 6144, 2048, 2560), (2560, 2048, 2048, 2560)]:
    x = (torch.randn(M, K, device="cuda") * 0.1).bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    whole = t(lambda: ops.gemm_nt(x, w, b, act=1, out=y))
    line = f"{M}x{N}x{K}: one call {whole:7.1f} us ({2.0 * M * N * K / whole / 1e6:6.0f} TF)"
    if m1 < M:
        a = t(lambda: ops.gemm_nt(x[:m1], w, b, act=1, out=y[:m1]))
        c = t(lambda: ops.gemm_nt(x[m1:], w, b, act=1, out=y[m1:]))
        both = t(lambda: (ops.gemm_nt(x[:m1], w, b, act=1, out=y[:m1]), ops.gemm_nt(x[m1:], w, b, act=1, out=y[m1:])))
        line += f" | rows [0,{m1}) {a:6.1f} us + rows [{m1},{M}) {c:6.1f} us; back to back {both:6.1f} us"
    print(line, flush=True)
