"""Per-workgroup timeline of the self-attention forward kernel (a -DSA32_TRACE=1 build): where does a workgroup's life go, and how
long does a CU slot stay empty between two workgroups?   MMGL_LIB_PATH=variants/lib_trace.so python tools/probes/sa32_trace.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
B, H, T, D = 64, 32, 640, 64
nwg = B * H * ((T + 127) // 128)
trace = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
os.environ["MMGL_SA32_TRACE"] = str(trace.data_ptr())
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402

L = _lib.lib()
d = H * D
q = (torch.randn(B, T, d, device="cuda") * 0.2).bfloat16()
k = torch.randn(B, T, d, device="cuda").bfloat16()
v = torch.randn(B, T, d, device="cuda").bfloat16()
valid = torch.ones(B, T, dtype=torch.uint8, device="cuda")
out = torch.empty_like(q)
lse = torch.empty(B, H, T, dtype=torch.float32, device="cuda")
for _ in range(3):
    assert L.mmgl_selfattn_fwd(ptr(q), ptr(k), ptr(v), ptr(valid), ptr(out), ptr(lse), B, H, T, D, 0, 1, stream_ptr()) == 0
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nwg, 8)
t0 = t[:, 0].min()
us = lambda x: (x - t0) / 100.0                    # 100 MHz wall clock
start, pro, first, loop_end, end = (us(t[:, i]) for i in range(5))
print(f"kernel span {end.max():.1f} us, {nwg} workgroups")
print(f"per workgroup (us): prologue {np.mean(pro - start):.2f}  first tile {np.mean(first - pro):.2f}  rest of loop {np.mean(loop_end - first):.2f}  "
      f"epilogue+store ack {np.mean(end - loop_end):.2f}  life {np.mean(end - start):.2f}")
for n in sorted(set(t[:, 7])):
    m = t[:, 7] == n
    print(f"  nkt {n:2d}: {m.sum():5d} wgs  prologue {np.mean((pro - start)[m]):5.2f}  first {np.mean((first - pro)[m]):5.2f}  loop {np.mean((loop_end - first)[m]):6.2f}  "
          f"({np.mean((loop_end - first)[m]) / max(n - 1, 1):.2f}/tile)  epilogue {np.mean((end - loop_end)[m]):5.2f}")
# CU slot occupancy: group by (xcc, se, sh, cu), sweep the busy intervals
hw, xcc = t[:, 5], t[:, 6] & 0xf
cu = ((xcc << 16) | (hw & 0xff00))                  # cu_id[11:8], sh_id[12], se_id[15:13]
conc = []
gaps = []
for c in np.unique(cu):
    m = cu == c
    s_, e_ = np.sort(start[m]), np.sort(end[m])
    busy = np.sum(end[m] - start[m])
    conc.append(busy / end.max())
print(f"{len(np.unique(cu))} CUs seen; average resident workgroups per CU over the kernel: {np.mean(conc):.2f} (min {np.min(conc):.2f}, max {np.max(conc):.2f})")
print(f"first start spread: {np.percentile(start, 50):.1f} us median, {np.percentile(start, 99):.1f} us p99 (launch ramp)")
