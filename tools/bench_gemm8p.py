#!/usr/bin/env python
"""Correctness + speed of the persistent ping-pong GEMM (mmgl_gemm_nt fast path) against torch (hipBLASLt) on the frozen
path's shapes.  Interleaved rounds in one process (cdna guide 5.4 rule 24), random data.   python tools/bench_gemm8p.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402

L = _lib.lib()


def gemm(x, W, b=None, resid=None, zmask=None, act=0, scale=1.0, y=None):
    from mmgl_amd import ops
    return ops.gemm_nt(x, W, b, resid, zmask, act=act, out_scale=scale, out=y)


def ref(x, W, b=None, resid=None, zmask=None, act=0, scale=1.0):
    v = x.float() @ W.float().t()
    if b is not None:
        v = v + b.float()
    v = v * scale
    if act == 1:
        v = torch.relu(v)
    elif act == 2:
        v = torch.nn.functional.gelu(v)
    elif act == 3:
        v = v * torch.sigmoid(1.702 * v)
    elif act == 4:
        v = torch.nn.functional.gelu(v, approximate="tanh")
    if zmask is not None:
        v = torch.where(zmask.float() > 0, v, torch.zeros_like(v))
    if resid is not None:
        v = v + resid.float()
    return v


def check(M, N, K, act=0, bias=True, resid=False, zmask=False, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16() if bias else None
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16() if resid else None
    z = torch.randn(M, N, device="cuda", generator=g).bfloat16() if zmask else None
    y = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.bfloat16)      # guard rows: must stay NaN
    gemm(x, W, b, r, z, act, scale, y=y[:M])
    torch.cuda.synchronize()
    want = ref(x, W, b, r, z, act, scale)
    got = y[:M].float()
    err = (got - want).abs().max().item()
    tol = 0.02 * want.abs().max().item() + 1e-2
    guard_ok = bool(torch.isnan(y[M:].float()).all())
    fast = L.mmgl_gemm_nt_fast(M, N, K, K, K, N, 1)
    ok = err <= tol and guard_ok and bool(torch.isfinite(got).all())
    print(f"check M={M:6d} N={N:6d} K={K:5d} act={act} bias={int(bias)} resid={int(resid)} zmask={int(zmask)} fast={fast} max err {err:.4f} (tol {tol:.4f}) guard {guard_ok} -> {'ok' if ok else 'FAIL'}",
          flush=True)
    return ok


def bench(M, N, K, act=0, rounds=5, iters=10):
    x = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K

    def t_ours():
        gemm(x, W, b, act=act, y=y)

    def t_old():
        _lib.check(L.mmgl_linear_fwd(ptr(x), ptr(W), ptr(b), ptr(y), M, N, K, act if act <= 1 else 0, 1.0, 1, stream_ptr()))

    def t_lib():
        if act == 1:
            torch._addmm_activation(b, x, W.t(), use_gelu=False)
        else:
            torch.nn.functional.linear(x, W, b)

    res = {"8p": [], "lib": []}
    fns = {"8p": t_ours, "lib": t_lib}
    for f in fns.values():
        for _ in range(3):
            f()
    for _ in range(rounds):
        for name, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            res[name].append(s.elapsed_time(e) / iters * 1e-3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    mn = {k: min(v) for k, v in res.items()}
    print(f"bench M={M:6d} N={N:6d} K={K:5d} act={act}: 8p {med['8p']*1e6:8.1f} us {fl/med['8p']/1e12:7.1f} TF (best {fl/mn['8p']/1e12:7.1f}) | "
          f"hipBLASLt {med['lib']*1e6:8.1f} us {fl/med['lib']/1e12:7.1f} TF (best {fl/mn['lib']/1e12:7.1f}) | ratio {med['lib']/med['8p']:.3f}", flush=True)


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = True
    ok &= check(4096, 2048, 2048)
    ok &= check(4096, 2048, 256, bias=False)
    ok &= check(40960, 2048, 2048, act=1)
    ok &= check(40000, 2000 // 16 * 16, 768, act=2)            # ragged M and N edges
    ok &= check(5000, 50272, 2048, bias=False)                # lm_head
    ok &= check(8192, 8192, 2048, act=3, resid=True)
    ok &= check(8192, 2048, 8192, zmask=True, bias=False, scale=0.5)
    ok &= check(4099, 3072, 768, act=4, resid=True, zmask=True)
    ok &= check(1024, 512, 256)                               # small: composed fallback
    ok &= check(2560, 2048, 2048, act=1)                      # few tiles (the reference's batch of 4): K-split work items
    ok &= check(2560, 2048, 8192, zmask=True, bias=False)
    ok &= check(2600, 2048, 6144, resid=True, act=3)          # uneven splits (48 K steps over 3), ragged M
    ok &= check(6500, 768, 3072, act=2)
    print("ALL OK" if ok else "SOME FAILED", flush=True)
    if not quick:
        for (M, N, K, act) in [(40960, 2048, 2048, 0), (40960, 6144, 2048, 0), (40960, 8192, 2048, 1), (40960, 2048, 8192, 0),
                               (40960, 8192, 2048, 0), (10240, 2048, 2048, 0), (10240, 8192, 2048, 1), (8192, 8192, 8192, 0),
                               (4096, 4096, 4096, 0), (100000, 2304, 768, 0), (100000, 3072, 768, 2), (100000, 768, 3072, 0),
                               (8192, 50272, 2048, 0)]:
            bench(M, N, K, act)
    sys.exit(0 if ok else 1)
