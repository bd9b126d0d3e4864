#!/usr/bin/env python
"""Outputs of the self-attention kernels (fwd, dq, dk, dv) on padded batches, saved for a bitwise comparison between two builds of the
library:   MMGL_LIB_PATH=variants/lib_old.so python tools/probes/sa32_bitwise.py /tmp/a.pt ; python tools/probes/sa32_bitwise.py /tmp/b.pt /tmp/a.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

CASES = [(3, 4, 640, 64, 0), (2, 2, 2176, 128, 0), (2, 3, 1000, 128, 0), (4, 32, 640, 64, 0), (3, 4, 640, 64, 20), (1, 2, 300, 128, 64), (2, 2, 100, 64, 0)]
out = {}
for ci, (B, H, T, D, P) in enumerate(CASES):
    g = torch.Generator().manual_seed(100 + ci)
    d = H * D
    q = (torch.randn(B, T, d, generator=g) * 0.3).bfloat16().cuda().requires_grad_()
    k = torch.randn(B, P + T, d, generator=g).bfloat16().cuda().requires_grad_()
    v = torch.randn(B, P + T, d, generator=g).bfloat16().cuda().requires_grad_()
    w = torch.randn(B, T, d, generator=g).bfloat16().cuda()
    am = torch.ones(B, P + T, dtype=torch.long)
    for b in range(B):                                   # WikiWeb2M layout: prompt | pad | summary | pad, ragged per sample
        lp = int(torch.randint(T // 10, T - T // 5, (1,), generator=g))
        ls = int(torch.randint(2, T // 10, (1,), generator=g))
        am[b, P + lp: P + T - T // 5] = 0
        am[b, P + T - T // 5 + ls:] = 0
    if B > 2:
        am[2, P + 70: P + T] = 0                          # nearly everything dead
    am = am.cuda()
    o = ops.selfattn_core_prefix(q, k, v, am, H, P) if P else ops.selfattn_core(q, k, v, am, H)
    gq, gk, gv = torch.autograd.grad((o * w).sum(), (q, k, v))
    out[ci] = [t.detach().cpu() for t in (o, gq, gk, gv)]
torch.save(out, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    ok = True
    for ci, ts in out.items():
        eq = [bool(torch.equal(a, b)) for a, b in zip(ts, ref[ci])]
        fin = all(bool(torch.isfinite(t.float()).all()) for t in ts)
        print(f"case {CASES[ci]}: out/dq/dk/dv bitwise equal: {eq}  finite: {fin}")
        ok &= all(eq) and fin
    print("ALL BITWISE EQUAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
