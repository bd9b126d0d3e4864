"""Times of the step's main GEMM shapes for one build of the library (MMGL_LIB_PATH); run once per variant, same box.
    MMGL_LIB_PATH=build_probe/libmmgl_X.so python tools/probes/gemm_variants.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402

SHAPES = [(40960, 8192, 2048, 1, True, False), (40960, 2048, 8192, 0, True, False), (40960, 6144, 2048, 0, True, False),
          (40960, 2048, 2048, 0, True, False), (40960, 8192, 2048, 0, False, True), (103777, 3072, 768, 2, True, False),
          (103777, 2304, 768, 0, True, False), (63040, 3072, 768, 3, True, False), (8192, 50272, 2048, 0, False, False)]
out = []
tot = 0.0
for (M, N, K, act, bias, zm) in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16() if bias else None
    z = torch.randn(M, N, device="cuda").bfloat16() if zm else None
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: ops.gemm_nt(x, w, b, None, z, act=act, out=y)
    for _ in range(3):
        f()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            f()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5 * 1e3)
    t = sorted(ts)[2]
    tot += t
    out.append(f"{t:7.1f}")
    del x, w, y, z
print(os.path.basename(os.environ.get("MMGL_LIB_PATH", "shipped")), " ".join(out), f"| sum {tot:8.1f} us", flush=True)
