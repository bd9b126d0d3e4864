#!/usr/bin/env python
"""Library dgrad GEMM of a frozen nn.Linear: dy @ W (NN) vs F.linear(dy, W^T.contiguous()) (NT) at the OPT-1.3B shapes."""
import torch
import torch.nn.functional as F


def t(fn, iters=30):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e-3


M = 10240
for N, K in [(2048, 2048), (6144, 2048), (8192, 2048), (2048, 8192), (50272, 2048)]:
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    Wt = W.t().contiguous()
    x = torch.randn(M, K, device="cuda").bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    fl = 2.0 * M * N * K
    a, b, c = t(lambda: F.linear(x, W)), t(lambda: dy @ W), t(lambda: F.linear(dy, Wt))
    print(f"N={N:6d} K={K:5d}  fwd {a*1e6:7.1f} us {fl/a/1e12:6.0f} TF | dgrad NN {b*1e6:7.1f} us {fl/b/1e12:6.0f} TF | dgrad NT(W^T copy) {c*1e6:7.1f} us {fl/c/1e12:6.0f} TF")
