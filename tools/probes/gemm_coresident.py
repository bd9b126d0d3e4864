#!/usr/bin/env python
"""How much does a co-resident kernel (a collective's 16-64 workgroups on a side stream) cost the persistent GEMM, whose grid is one
workgroup per CU with the whole LDS / register file of its CU?   python tools/probes/gemm_coresident.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mmgl_amd import ops  # noqa: E402

so = os.path.join(ROOT, "variants", "liboccupier.so")      # hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o variants/liboccupier.so tools/probes/occupier.hip
occ = ctypes.CDLL(so if os.path.exists(so) else os.path.join(ROOT, "build_probe", "liboccupier.so"))
occ.occupier_spin.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]


def main():
    # 5 rounds of tiles (the round-3 measurement), then the few-round outputs of ADVICE round 3: 1 / 1.5 / 2 / 2.5 rounds
    for M in [int(a) for a in sys.argv[1:]] or [40960, 8192, 12288, 16384, 20480]:
        one(M, 2048, 2048)


def one(M, N, K):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    for _ in range(5):
        ops.gemm_nt(x, w, out=y)
    torch.cuda.synchronize()
    for dynamic in (False, True):
      ops.gemm_dynamic_schedule(dynamic)
      for wgs in (0, 8, 16, 32, 64):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            if wgs:
                occ.occupier_spin(wgs, 100 * 1000 * 6, side.cuda_stream)      # 6 ms of the 100 MHz wall clock: covers the loop below
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.gemm_nt(x, w, out=y)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 8 * 1e3)
        print(f"{'dynamic' if dynamic else 'static ':7s} schedule, occupier workgroups {wgs:3d}: {M}x{N}x{K} {best:7.1f} us per GEMM", flush=True)
    ops.gemm_dynamic_schedule(False)


if __name__ == "__main__":
    main()
