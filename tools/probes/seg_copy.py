#!/usr/bin/env python
"""What the cross-attention core's access pattern costs by itself (tools/probes/seg_copy.hip): Q -> O copy of [64, 640, 32 * 64] bf16 with
1 / 2 / 4 / 8 heads (128 B .. 1 KiB per row) per workgroup, against torch's linear copy of the same bytes.   python tools/probes/seg_copy.py"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = os.path.join(ROOT, "variants", "libsegcopy.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "probes", "seg_copy.hip")], check=True)
lib = ctypes.CDLL(so)
lib.seg_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]
B, T, H, D = 64, 640, 32, 64
x = torch.randn(B, T, H * D, device="cuda").bfloat16()
y = torch.empty_like(x)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nbytes = 2 * x.numel() * 2


def t(fn, n=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


us = t(lambda: y.copy_(x))
print(f"torch copy_ (linear): {us:6.1f} us = {nbytes / us / 1e6:5.2f} TB/s", flush=True)
for heads in (1, 2, 4, 8):
    for nv in (1, 2, 4, 8):
        y.zero_()
        assert lib.seg_copy(x.data_ptr(), y.data_ptr(), B, T, H, D, heads, nv, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(x, y), (heads, nv)
        us = t(lambda: lib.seg_copy(x.data_ptr(), y.data_ptr(), B, T, H, D, heads, nv, st))
        print(f"heads per workgroup {heads} ({heads * D * 2:4d} B per row), {nv} x 16 B per lane and step ({B * H // heads} workgroups): {us:6.1f} us = {nbytes / us / 1e6:5.2f} TB/s", flush=True)
