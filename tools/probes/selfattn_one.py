"""A few launches of the causal self-attention kernels at one batch size (target of tools/pmc_sq.sh).   python tools/probes/selfattn_one.py <B> [T] [D]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_selfattn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 640
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
bench_selfattn.run(B, H=2048 // D if D == 64 else 32, T=T, D=D, iters=5)
