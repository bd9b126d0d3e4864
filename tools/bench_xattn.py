#!/usr/bin/env python
"""Micro-benchmark of the cross-attention core through the C ABI: raw ctypes launches (pre-allocated outputs, no
autograd) bracketed by HIP events; GB/s against the algorithmic bytes of SURVEY.md 8(d).  Launch overhead of the
ctypes call (~3 us) bounds tiny shapes: for kernel-only time run it under `rocprofv3 --kernel-trace --stats`.
    python tools/bench_xattn.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402


def run(B, H=32, T=640, S=64, D=64, dtype=torch.bfloat16, iters=200):
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    d = H * D
    q = (torch.randn(B, T, d, generator=g) * 0.2).to(dtype).cuda()
    k = torch.randn(B, S, d, generator=g).to(dtype).cuda()
    v = torch.randn(B, S, d, generator=g).to(dtype).cuda()
    w = torch.randn(B, T, d, generator=g).to(dtype).cuda()
    valid = torch.rand(B, S, generator=g) > 0.3
    valid[:, 0] = True
    sv = valid.sum(1).tolist()
    valid = valid.to(torch.uint8).cuda()
    e = q.element_size()
    fb = sum(2.0 * T * d * e + 2.0 * s * d * e for s in sv)
    bb = sum(3.0 * T * d * e + 4.0 * s * d * e for s in sv)
    out = torch.empty_like(q)
    lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    nws = L.mmgl_xattn_bwd_workspace(B, H, T, S, D)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    code = _lib.dtype_code(q)
    st = stream_ptr()
    fwd = lambda: L.mmgl_xattn_fwd(ptr(q), ptr(k), ptr(v), ptr(valid), ptr(out), ptr(lse), B, H, T, S, D, code, st)
    bwd = lambda: L.mmgl_xattn_bwd(ptr(w), ptr(q), ptr(k), ptr(v), ptr(lse), ptr(valid), ptr(dq), ptr(dk), ptr(dv), ptr(ws), nws, B, H, T, S, D, code, st)
    for _ in range(5):
        assert fwd() == 0 and bwd() == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        fwd()
    ev[1].record()
    ev[2].record()
    for _ in range(iters):
        bwd()
    ev[3].record()
    torch.cuda.synchronize()
    tf = ev[0].elapsed_time(ev[1]) / iters * 1e-3
    tb = ev[2].elapsed_time(ev[3]) / iters * 1e-3
    print(f"B={B:3d} H={H} T={T} S={S} D={D} {str(dtype)[6:]:9s} fwd {tf*1e6:7.1f} us {fb/tf/1e9:7.0f} GB/s ({fb/tf/8e12*100:4.1f}% of 8 TB/s) | "
          f"bwd {tb*1e6:7.1f} us {bb/tb/1e9:7.0f} GB/s ({bb/tb/8e12*100:4.1f}%)", flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["llama"]:                  # config 5's shape (D = 128, S = 128, T = 2176) at the bench batch and around it
        for B in (8, 4, 16):
            run(B, H=32, T=2176, S=128, D=128)
        run(64, H=32, T=640, S=64, D=64)
        sys.exit(0)
    Bs = [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32]
    for B in Bs:
        run(B)
    if len(sys.argv) <= 1:
        run(4, H=12, S=16)
        run(2, H=32, T=2176, S=128, D=128)
        run(8, dtype=torch.float32)
