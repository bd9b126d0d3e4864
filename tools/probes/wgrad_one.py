"""A few weight-gradient launches (mmgl_linear_bwd, dW only) at one shape: the target of tools/pmc_sq.sh for gemm8p_tt_kernel.
python tools/probes/wgrad_one.py M N K"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402

L = _lib.lib()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (40960, 8192, 2048)
x = torch.randn(M, K, device="cuda").bfloat16()
W = torch.randn(N, K, device="cuda").bfloat16()
dy = (torch.randn(M, N, device="cuda") * 0.1).bfloat16()
dW = torch.empty_like(W)
nws = L.mmgl_linear_bwd_workspace(M, N, K, 0, 1)
ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
for _ in range(6):
    _lib.check(L.mmgl_linear_bwd(ptr(dy), None, ptr(x), ptr(W), None, ptr(dW), None, ptr(ws), nws, M, N, K, 0, 1.0, 0, 0, 1, stream_ptr()))
torch.cuda.synchronize()
