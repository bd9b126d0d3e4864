"""Clock stamps of workgroup 0 of the half-tile GEMM (H8_TRACE build): one stamp per cycle (8 phases) of waves 0 (group A) and 4
(group B).  A half-tile of K = 64 nk is nk / 2 compute cycles + 1 epilogue cycle; B starts one cycle after A.
    MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_trace.so python tools/probes/gemm8h_trace.py [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
buf = torch.zeros(8 * 128, dtype=torch.int64, device="cuda")
os.environ["MMGL_H8_TRACE"] = hex(buf.data_ptr())
from mmgl_amd import ops  # noqa: E402

M, N = 40960, 2048
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    buf.zero_()
    ops.gemm_nt(x, w, out=y)
torch.cuda.synchronize()
t = buf.view(8, 128).cpu()
C = K // 128
P = C + 1
for wv, off in ((0, 0), (4, 1)):
    row = [v for v in t[wv].tolist() if v]
    d = [b - a for a, b in zip(row, row[1:])]
    tags = []
    for c in range(len(d)):
        loc = c - off
        tags.append("idle" if loc < 0 else ("EPI" if loc % P == C else ("c0" if loc % P == 0 else "c")))
    print(f"wave {wv}: {len(row)} stamps; cycle durations (clocks):")
    print("   " + " ".join(f"{tg}:{v}" for tg, v in zip(tags, d)))
    comp = [v for tg, v in zip(tags, d) if tg == "c"]
    epi = [v for tg, v in zip(tags, d) if tg == "EPI"]
    c0 = [v for tg, v in zip(tags, d) if tg == "c0"]
    if comp:
        print(f"   compute cycle mean {sum(comp) / len(comp):.0f}, first-of-tile {sum(c0) / max(1, len(c0)):.0f}, epilogue cycle mean {sum(epi) / max(1, len(epi)):.0f}")
