#!/usr/bin/env python
"""Micro-benchmark of the MFMA linears through the C ABI (raw ctypes launches, HIP events).  python tools/bench_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402


def run(M, N, K, act=0, dtype=torch.bfloat16, iters=30):
    L = _lib.lib()
    x = torch.randn(M, K, device="cuda").to(dtype)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
    b = torch.randn(N, device="cuda").to(dtype)
    y = torch.empty(M, N, device="cuda", dtype=dtype)
    dy = torch.randn(M, N, device="cuda").to(dtype)
    dx, dW, db = torch.empty_like(x), torch.empty_like(W), torch.empty_like(b)
    code = _lib.dtype_code(x)
    nws = L.mmgl_linear_bwd_workspace(M, N, K, act, code)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    st = stream_ptr()
    f = lambda: L.mmgl_linear_fwd(ptr(x), ptr(W), ptr(b), ptr(y), M, N, K, act, 1.0, code, st)
    g = lambda: L.mmgl_linear_bwd(ptr(dy), ptr(y), ptr(x), ptr(W), ptr(dx), ptr(dW), ptr(db), ptr(ws), nws, M, N, K, act, 1.0, 0, 0, code, st)
    for _ in range(3):
        assert f() == 0 and g() == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        f()
    ev[1].record()
    ev[2].record()
    for _ in range(iters):
        g()
    ev[3].record()
    torch.cuda.synchronize()
    tf = ev[0].elapsed_time(ev[1]) / iters * 1e-3
    tb = ev[2].elapsed_time(ev[3]) / iters * 1e-3
    fl = 2.0 * M * N * K
    # hipBLASLt reference for the same forward (torch F.linear)
    torch.nn.functional.linear(x, W, b)
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev2[0].record()
    for _ in range(iters):
        torch.nn.functional.linear(x, W, b)
    ev2[1].record()
    torch.cuda.synchronize()
    tt = ev2[0].elapsed_time(ev2[1]) / iters * 1e-3
    print(f"M={M:6d} N={N:5d} K={K:5d} act={act} {str(dtype)[6:]:9s} fwd {tf*1e6:8.1f} us {fl/tf/1e12:7.1f} TF | bwd(dx+dW+db) {tb*1e6:8.1f} us {2*fl/tb/1e12:7.1f} TF"
          f" | torch F.linear fwd {tt*1e6:8.1f} us {fl/tt/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        M = int(sys.argv[1])
        run(M, 2048, 2048)
        run(M, 8192, 2048, act=1)
        run(M, 2048, 8192)
        sys.exit(0)
    for M in (5120, 10240):
        run(M, 2048, 2048)
        run(M, 8192, 2048, act=1)
        run(M, 2048, 8192)
    run(1024, 2048, 2048)
    run(4096, 4096, 4096)
    run(8192, 8192, 8192, iters=10)
    run(5120, 2048, 2048, dtype=torch.float32, iters=10)
