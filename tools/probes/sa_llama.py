#!/usr/bin/env python
"""Causal self-attention at config 5's shape (H = 32, T = 2176, D = 128) as a function of the batch: fwd / bwd times by HIP events.
    python tools/probes/sa_llama.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import bench_selfattn  # noqa: E402

for B in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 12, 16, 32]:
    bench_selfattn.run(B, H=32, T=2176, D=128, masked=False, iters=20)
