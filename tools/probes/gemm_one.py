#!/usr/bin/env python
"""A few launches of the forward / backward linears at one shape (target of tools/pmc_sq.sh).  args: M N K [act]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_gemm  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4])
bench_gemm.run(M, N, K, act=int(sys.argv[4]) if len(sys.argv) > 4 else 0, iters=5)
