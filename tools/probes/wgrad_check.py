"""Weight gradient through mmgl_linear_bwd at the cross-attention layers' shapes: parity vs fp32 torch + TF (dW only and dx + dW + db)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import _lib  # noqa: E402
from mmgl_amd._lib import ptr, stream_ptr  # noqa: E402

L = _lib.lib()


def run(M, N, K, check=True):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    dy = (torch.randn(M, N, device="cuda", generator=g) * 0.1).bfloat16()
    dW = torch.empty_like(W)
    db = torch.empty(N, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    nws = L.mmgl_linear_bwd_workspace(M, N, K, 0, 1)
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
    f_w = lambda: _lib.check(L.mmgl_linear_bwd(ptr(dy), None, ptr(x), ptr(W), None, ptr(dW), None, ptr(ws), nws, M, N, K, 0, 1.0, 0, 0, 1, stream_ptr()))
    f_all = lambda: _lib.check(L.mmgl_linear_bwd(ptr(dy), None, ptr(x), ptr(W), ptr(dx), ptr(dW), ptr(db), ptr(ws), nws, M, N, K, 0, 1.0, 0, 0, 1, stream_ptr()))
    f_w()
    torch.cuda.synchronize()
    if check:
        want = dy.float().t() @ x.float()
        err = (dW.float() - want).abs().max().item()
        tol = 2e-2 * want.abs().max().item() + 1e-3
        print(f"  dW max err {err:.4f} (tol {tol:.4f}) -> {'ok' if err <= tol else 'FAIL'}", flush=True)
    for name, f, fl in (("dW", f_w, 2.0 * M * N * K), ("dx+dW+db", f_all, 4.0 * M * N * K)):
        for _ in range(2):
            f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 10 * 1e-3
        print(f"M={M} N={N} K={K} {name}: {t*1e6:.1f} us {fl/t/1e12:.1f} TF", flush=True)


if len(sys.argv) > 1:        # python tools/probes/wgrad_check.py M,N,K [M,N,K ...]
    for arg in sys.argv[1:]:
        run(*(int(v) for v in arg.split(",")), check=False)
    sys.exit(0)
for shape in [(40960, 2048, 2048), (40960, 8192, 2048), (40960, 2048, 8192), (4096, 2048, 2048), (17408, 4096, 4096), (17408, 11008, 4096)]:
    run(*shape)
