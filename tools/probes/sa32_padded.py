#!/usr/bin/env python
"""Self-attention kernels on WikiWeb2M-shaped padding (bench.py's synthetic lengths: prompt U{64..Lin} of Lin, summary U{8..64} of 128) and dense,
config 3's and config 5's shapes.   python tools/probes/sa32_padded.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402


def run(B, H, T, D, lin, padded, iters=30):
    L = _lib.lib()
    d = H * D
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(B, T, d, generator=g) * 0.2).bfloat16().cuda()
    k, v, w = (torch.randn(B, T, d, generator=g).bfloat16().cuda() for _ in range(3))
    valid = torch.ones(B, T, dtype=torch.uint8)
    if padded:
        for b in range(B):
            lp = int(torch.randint(64, lin + 1, (1,), generator=g)); ls = int(torch.randint(8, 65, (1,), generator=g))
            valid[b, lp:lin] = 0; valid[b, lin + ls:] = 0
    valid = valid.cuda()
    out = torch.empty_like(q); lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    nws = L.mmgl_selfattn_bwd_workspace(B, H, T); ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    code = _lib.dtype_code(q); st = stream_ptr()
    fwd = lambda: L.mmgl_selfattn_fwd(ptr(q), ptr(k), ptr(v), ptr(valid), ptr(out), ptr(lse), B, H, T, D, 0, code, st)
    bwd = lambda: L.mmgl_selfattn_bwd(ptr(w), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(valid), ptr(dq), ptr(dk), ptr(dv), ptr(ws), nws, B, H, T, D, 0, 0, code, st)
    for _ in range(3):
        assert fwd() == 0 and bwd() == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        fwd()
    ev[1].record(); ev[2].record()
    for _ in range(iters):
        bwd()
    ev[3].record()
    torch.cuda.synchronize()
    tf, tb = ev[0].elapsed_time(ev[1]) / iters * 1e-3, ev[2].elapsed_time(ev[3]) / iters * 1e-3
    fl = 4.0 * B * T * T * d / 2
    print(f"B={B:3d} H={H} T={T} D={D} {'padded' if padded else 'dense '} fwd {tf*1e6:8.1f} us {fl/tf/2.5e15:6.3f} of peak (nominal causal) | bwd {tb*1e6:8.1f} us {2.5*fl/tb/2.5e15:6.3f}", flush=True)


for padded in (True, False):
    run(64, 32, 640, 64, 512, padded)
    run(8, 32, 2176, 128, 2048, padded)
