#!/usr/bin/env python
"""Micro-benchmark of the cross-attention core through the C ABI (HIP events, many launches): GB/s against the
algorithmic bytes of SURVEY.md 8(d).   python tools/bench_xattn.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgl_amd import ops  # noqa: E402


def run(B, H=32, T=640, S=64, D=64, dtype=torch.bfloat16, iters=50):
    g = torch.Generator().manual_seed(0)
    d = H * D
    q = (torch.randn(B, T, d, generator=g) * 0.2).to(dtype).cuda().requires_grad_()
    k = torch.randn(B, S, d, generator=g).to(dtype).cuda().requires_grad_()
    v = torch.randn(B, S, d, generator=g).to(dtype).cuda().requires_grad_()
    w = torch.randn(B, T, d, generator=g).to(dtype).cuda()
    valid = torch.rand(B, S, generator=g) > 0.3
    valid[:, 0] = True
    sv = valid.sum(1).tolist()
    valid = valid.cuda()
    e = q.element_size()
    fb = sum(2.0 * T * d * e + 2.0 * s * d * e for s in sv)
    bb = sum(3.0 * T * d * e + 4.0 * s * d * e for s in sv)
    for _ in range(3):
        o = ops.xattn_core(q, k, v, valid, H)
        o.backward(w)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.no_grad():
        ev[0].record()
        for _ in range(iters):
            ops.xattn_core(q, k, v, valid, H)
        ev[1].record()
    o = ops.xattn_core(q, k, v, valid, H)
    ev[2].record()
    for _ in range(iters):
        torch.autograd.grad(o, (q, k, v), w, retain_graph=True)
    ev[3].record()
    torch.cuda.synchronize()
    tf = ev[0].elapsed_time(ev[1]) / iters * 1e-3
    tb = ev[2].elapsed_time(ev[3]) / iters * 1e-3
    print(f"B={B:3d} H={H} T={T} S={S} D={D} {str(dtype)[6:]:9s} fwd {tf*1e6:7.1f} us {fb/tf/1e9:7.0f} GB/s ({fb/tf/8e12*100:4.1f}% of 8 TB/s) | "
          f"bwd {tb*1e6:7.1f} us {bb/tb/1e9:7.0f} GB/s ({bb/tb/8e12*100:4.1f}%)", flush=True)


if __name__ == "__main__":
    Bs = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32]
    for B in Bs:
        run(B)
    run(4, H=12, S=16)
    run(2, H=32, T=2176, S=128, D=128)
    run(8, dtype=torch.float32)
