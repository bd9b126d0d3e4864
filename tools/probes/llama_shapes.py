#!/usr/bin/env python
"""Attention kernels at the config-5 shapes (Llama-2-7B: H=32, D=128, T=2176, 32 neighbors x 4 tokens = 128 keys)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_selfattn  # noqa: E402
import bench_xattn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bench_xattn.run(B, H=32, T=2176, S=128, D=128, iters=20)
bench_selfattn.run(B, H=32, T=2176, D=128, iters=10)
