#!/usr/bin/env python
"""Streaming reference for the cross-attention core: a plain device copy of the same Q -> O bytes (read T*d, write T*d per
sample), i.e. what a kernel with zero compute and perfectly linear access gets on this part."""
import sys

import torch

for B in [int(a) for a in sys.argv[1:]] or [16, 48]:
    q = torch.randn(B, 640, 2048, device="cuda").bfloat16()
    o = torch.empty_like(q)
    for _ in range(3):
        o.copy_(q)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(50):
        o.copy_(q)
    ev[1].record()
    torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) / 50 * 1e-3
    nbytes = 2.0 * q.numel() * 2
    print(f"B={B:3d}  copy {nbytes/1e6:7.1f} MB in {t*1e6:7.1f} us = {nbytes/t/1e9:7.1f} GB/s")
