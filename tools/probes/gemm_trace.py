"""Clock stamps of workgroup 0 of the persistent GEMM (P8_TRACE build): where the time between output tiles goes.
    MMGL_LIB_PATH=build_probe/libmmgl_trace.so python tools/probes/gemm_trace.py [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
buf = torch.zeros(8 * 64, dtype=torch.int64, device="cuda")
os.environ["MMGL_P8_TRACE"] = hex(buf.data_ptr())
from mmgl_amd import ops  # noqa: E402

M, N = 40960, 8192
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm_nt(x, w, out=y)
torch.cuda.synchronize()
t = buf.view(8, 64).cpu()
labels = ["tile start"] + [f"kt0 ph{i}" for i in range(1, 9)] + [f"kt2 ph{i}" for i in range(1, 9)] + ["loop end", "epilogue done"]
for wv in (0, 4):
    row = t[wv].tolist()
    base = row[0]
    print(f"wave {wv}:")
    prev = base
    for i, v in enumerate(row):
        if v == 0:
            break
        lab = labels[i % len(labels)]
        print(f"   {lab:14s} {v - base:8d}  (+{v - prev})")
        prev = v
