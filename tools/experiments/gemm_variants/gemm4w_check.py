#!/usr/bin/env python
"""Four-wave persistent GEMM (gemm4w.hip, MMGL_GEMM_4W=1) against torch and against the eight-wave kernel: correctness on ragged shapes,
then timing, alternating the two kernels in one process (two libraries: MMGL_LIB_PATH is read once, so the eight-wave numbers come from
a child process).   python tools/probes/gemm4w_check.py [time]"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

SHAPES = [(40960, 2048, 2048), (40960, 6144, 2048), (40960, 2048, 8192), (43520, 2048, 2048), (16384, 2048, 2048), (40960, 2048, 768),
          (40960, 8192, 2048), (8192, 8192, 8192)]


def check(M, N, K, bias=True, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16() if bias else None
    y = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(x, w, b, None, None, act=0, out_scale=scale, out=y[:M])
    torch.cuda.synchronize()
    want = x.float() @ w.float().t()
    if b is not None:
        want = want + b.float()
    want = want * scale
    got = y[:M].float()
    err = (got - want).abs().max().item()
    tol = 0.02 * want.abs().max().item() + 1e-2
    guard = bool(torch.isnan(y[M:].float()).all())
    ok = err <= tol and guard and bool(torch.isfinite(got).all())
    nbad = int(((got - want).abs() > tol).sum()) + int((~torch.isfinite(got)).sum())
    print(f"check M={M:6d} N={N:6d} K={K:5d} bias={int(bias)} scale={scale} max err {err:.4f} (tol {tol:.4f}) bad {nbad} guard {guard} -> {'ok' if ok else 'FAIL'}", flush=True)
    if not ok and nbad:
        bad = ((got - want).abs() > tol) | ~torch.isfinite(got)
        idx = bad.nonzero()[:6].tolist()
        print("   first bad:", [(i, j, round(got[i, j].item(), 3), round(want[i, j].item(), 3)) for i, j in idx], flush=True)
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print(f"   bad rows {rows.numel()} [{rows[:8].tolist()} ..], bad cols {cols.numel()} [{cols[:8].tolist()} ..]", flush=True)
    return ok


def timing():
    tag = "4w" if os.environ.get("MMGL_GEMM_4W") else "8p"
    out = []
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm_nt(x, w, b, None, None, act=0, out=y)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_nt(x, w, b, None, None, act=0, out=y)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        out.append(f"{M}x{N}x{K}: {best:7.1f} us {2.0 * M * N * K / best / 1e6:6.0f} TF")
    print(tag, "|", " | ".join(out), flush=True)


if __name__ == "__main__":
    if "time" in sys.argv:
        timing()
    else:
        ok = True
        for M, N, K in [(512, 512, 512), (256, 256, 1024), (1024, 768, 512), (8192, 2048, 2048), (40960, 2048, 2048), (4096, 2048, 8192)]:
            ok &= check(M, N, K)
        ok &= check(8192, 2048, 768, bias=False, scale=0.125)
        ok &= check(2000, 1040, 576)                      # ragged rows / columns
        ok &= check(8192, 50272, 2048)                    # lm_head: a 96-column last tile column
        ok &= check(300, 272, 512, bias=False)
        print("ALL OK" if ok else "SOME FAILED")
        sys.exit(0 if ok else 1)
