#!/usr/bin/env python
"""Library GEMM with a GELU epilogue (torch._addmm_activation) vs GEMM + one-pass activation kernel at the RoBERTa FFN shape."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402


def t(fn, iters=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3


M, K, N = 100000, 768, 3072
x = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
a = torch._addmm_activation(b, x, W.t(), use_gelu=True)
ref_erf = F.gelu(F.linear(x.float(), W.float(), b.float()))
ref_tanh = F.gelu(F.linear(x.float(), W.float(), b.float()), approximate="tanh")
mine = ops.activation_(F.linear(x, W, b), "gelu")
print("epilogue vs erf  max abs", (a.float() - ref_erf).abs().max().item(), " vs tanh", (a.float() - ref_tanh).abs().max().item())
print("own kernel vs erf max abs", (mine.float() - ref_erf).abs().max().item())
print("us: linear+act kernel", t(lambda: ops.activation_(F.linear(x, W, b), "gelu")), " epilogue", t(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=True)),
      " linear only", t(lambda: F.linear(x, W, b)))
