#!/usr/bin/env python
"""Shader clock under load (tools/probes/clock_probe.hip): an MFMA loop on every CU against a VALU loop, then the clock the persistent
GEMM itself runs at (kernel duration x MFMA count -> cycles per MFMA).   python tools/probes/clock_probe.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = ctypes.CDLL(os.path.join(ROOT, "variants", "libclockprobe.so"))
lib.clock_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(2 * 256, dtype=torch.int64, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode, name, iters in ((0, "VALU fma loop ", 2_000_000), (1, "MFMA 16x16x32  ", 400_000), (1, "MFMA 16x16x32  ", 1_600_000)):
    for wgs in (256, 32):
        lib.clock_probe(wgs, iters, mode, ctypes.c_void_p(out.data_ptr()), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.clock_probe(wgs, iters, mode, ctypes.c_void_p(out.data_ptr()), st)
        e1.record()
        torch.cuda.synchronize()
        o = out[:2 * wgs].view(wgs, 2).double()
        ghz = (o[:, 0] / o[:, 1]).mean().item() * 0.1
        ms = e0.elapsed_time(e1)
        extra = ""
        if mode == 1:
            n = 4.0 * iters * 8 * wgs                                  # MFMAs: 4 per iteration and wave, 8 waves per workgroup
            tf = n * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12
            extra = f"  {tf:7.1f} TF  ({n / (wgs * 4) * 16 / (ms * 1e-3) / 1e9:.2f} G MFMA-cycles/s per SIMD at 16 cycles each)"
        print(f"{name} {wgs:3d} workgroups x 8 waves, {ms:8.2f} ms: shader clock {ghz:.3f} GHz (s_memtime / s_memrealtime){extra}", flush=True)

# ---- the shader clock WHILE the persistent GEMM runs: a one-workgroup sampler on a side stream next to 40960x8192x2048 GEMMs
from mmgl_amd import ops  # noqa: E402

M, N, K = 40960, 8192, 2048
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_nt(x, w, out=y)
torch.cuda.synchronize()
side = torch.cuda.Stream()
sst = ctypes.c_void_p(side.cuda_stream)
samples = torch.zeros(16, 2, dtype=torch.int64, device="cuda")
ops.gemm_dynamic_schedule(True)              # work-stealing tiles: the sampler's CU does not hold the GEMM back
for label, run_gemm in (("idle chip", False), ("during GEMMs", True), ("during GEMMs", True), ("idle chip", False)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    if run_gemm:
        for _ in range(60):
            ops.gemm_nt(x, w, out=y)
    e1.record()
    for i in range(12):
        lib.clock_probe(1, 150_000, 0, ctypes.c_void_p(samples[i].data_ptr()), sst)
    torch.cuda.synchronize()
    s = samples[:12].double().cpu()
    ghz = (s[:, 0] / s[:, 1] * 0.1).tolist()
    msg = f"{label:13s}: sampler windows of {s[0, 1].item() / 100:.0f} us -> shader clock " + " ".join(f"{g:.2f}" for g in ghz) + " GHz"
    if run_gemm:
        ms = e0.elapsed_time(e1) / 60
        msg += f"   (GEMM {ms * 1e3:.0f} us = {2.0 * M * N * K / ms / 1e9:.0f} TF)"
    print(msg, flush=True)
