"""Does the output row pitch matter to the persistent GEMM's per-tile store cost?  Same GEMM, output written into column slices
of wider buffers (pitch = N, N + 64, N + 128, 2N elements).   python tools/probes/gemm_pitch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mmgl_amd import ops  # noqa: E402

for (M, N, K) in [(40960, 8192, 2048), (40960, 2048, 2048), (103777, 3072, 768)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    for pitch in (N, N + 64, N + 128, N + 1024, 2 * N):
        buf = torch.empty(M, pitch, device="cuda", dtype=torch.bfloat16)
        y = buf[:, :N]
        for _ in range(3):
            ops.gemm_nt(x, w, out=y)
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                ops.gemm_nt(x, w, out=y)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 5 * 1e3)
        t = sorted(ts)[2]
        print(f"M={M} N={N} K={K} pitch {pitch:6d}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF", flush=True)
        del buf, y
