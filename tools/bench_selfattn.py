#!/usr/bin/env python
"""Micro-benchmark of the causal self-attention kernels through the C ABI (raw ctypes launches, HIP events).
    python tools/bench_selfattn.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402


def run(B, H=32, T=640, D=64, dtype=torch.bfloat16, iters=50, masked=True):
    L = _lib.lib()
    d = H * D
    q = (torch.randn(B, T, d, device="cuda") * 0.2).to(dtype)
    k = torch.randn(B, T, d, device="cuda").to(dtype)
    v = torch.randn(B, T, d, device="cuda").to(dtype)
    w = torch.randn(B, T, d, device="cuda").to(dtype)
    valid = torch.ones(B, T, dtype=torch.uint8, device="cuda")
    if masked:
        valid[:, 400:512] = 0
    out = torch.empty_like(q)
    lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    nws = L.mmgl_selfattn_bwd_workspace(B, H, T)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    code = _lib.dtype_code(q)
    st = stream_ptr()
    fwd = lambda: L.mmgl_selfattn_fwd(ptr(q), ptr(k), ptr(v), ptr(valid), ptr(out), ptr(lse), B, H, T, D, 0, code, st)
    bwd = lambda: L.mmgl_selfattn_bwd(ptr(w), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(valid), ptr(dq), ptr(dk), ptr(dv), ptr(ws), nws, B, H, T, D, 0, 0, code, st)
    for _ in range(3):
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
    fl = 4.0 * B * T * T * d / 2            # causal half
    print(f"B={B:3d} H={H} T={T} D={D} {str(dtype)[6:]:9s} {'masked' if masked else 'dense ':6s} fwd {tf*1e6:8.1f} us {fl/tf/1e12:6.1f} TF(causal) | bwd {tb*1e6:8.1f} us {2.5*fl/tb/1e12:6.1f} TF", flush=True)


if __name__ == "__main__":
    for B in [int(a) for a in sys.argv[1:]] or [8, 16]:
        run(B)
        run(B, masked=False)
    if os.environ.get("BENCH_SA_LLAMA"):
        run(8, H=32, T=2176, D=128, masked=False)
