"""Does the ROW PITCH of the GEMM inputs matter (power-of-two pitches put every row of a tile's K slice on the same L2 channel)?
Interleaved A/B of x / W pitches K, K + 64, K + 128 elements.   python tools/probes/gemm_inpitch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402

for (M, N, K) in [(2560, 2048, 2048), (2560, 8192, 2048), (40960, 2048, 2048), (40960, 8192, 2048), (40960, 2048, 8192), (40960, 6144, 2048)]:
    res = {}
    ops_ = {}
    for pad in (0, 64, 128, 192):
        xb = torch.randn(M, K + pad, device="cuda").bfloat16()
        wb = (torch.randn(N, K + pad, device="cuda") * K ** -0.5).bfloat16()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops_[pad] = (xb[:, :K], wb[:, :K], y)
        res[pad] = []
        for _ in range(3):
            ops.gemm_nt(xb[:, :K], wb[:, :K], out=y)
    for _ in range(5):
        for pad, (x, w, y) in ops_.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                ops.gemm_nt(x, w, out=y)
            e.record()
            torch.cuda.synchronize()
            res[pad].append(s.elapsed_time(e) / 5 * 1e3)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"pitch K+{p}: {sorted(v)[len(v) // 2]:.1f} us ({2.0 * M * N * K / sorted(v)[len(v) // 2] / 1e6:.0f} TF)" for p, v in res.items()), flush=True)
