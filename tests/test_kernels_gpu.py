"""-m gpu parity tests for the row / elementwise / GEMM kernels (through the C ABI) against plain fp32 torch on CPU."""
import pytest
import torch
import torch.nn.functional as F

from helpers import Fixture, assert_close

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype, f32=1e-4, bf=2e-2):
    return f32 if dtype == torch.float32 else bf


def dev(t, dtype):
    return t.to(dtype).cuda().requires_grad_()


@pytest.mark.parametrize("rows,cols", [(37, 64), (640, 768), (2560, 2048), (100, 4096), (9, 1000)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm(rows, cols, dtype):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.5
    gamma = torch.randn(cols, generator=g) * 0.2 + 1
    beta = torch.randn(cols, generator=g) * 0.1
    w = torch.randn(rows, cols, generator=g)
    xd, gd, bd = dev(x, dtype), dev(gamma, dtype), dev(beta, dtype)
    y = ops.layer_norm(xd, gd, bd, 1e-5)
    (y * w.to(dtype).cuda()).sum().backward()
    xr, gr, br = (t.detach().float().cpu().requires_grad_() for t in (xd, gd, bd))
    yr = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    (yr * w.to(dtype).float()).sum().backward()
    t = tol(dtype)
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert_close(gd.grad.float(), gr.grad, t, "dgamma")
    assert_close(bd.grad.float(), br.grad, t, "dbeta")


@pytest.mark.parametrize("rows,cols", [(37, 64), (640, 2048), (9, 1000)])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_sum,affine_grad", [(True, True), (True, False), (False, False)])
def test_add_layer_norm_pair(rows, cols, dtype, use_sum, affine_grad):
    """(s, y) = (x + r, LN(x + r)) with both outputs feeding the loss: dx = dr = LN'(dy) + ds from one backward kernel."""
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(rows * 3 + cols)
    x, r = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g) * 2
    gamma, beta = torch.randn(cols, generator=g) * 0.2 + 1, torch.randn(cols, generator=g) * 0.1
    w1, w2 = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    xd, rd = dev(x, dtype), dev(r, dtype)
    gd, bd = (dev(gamma, dtype), dev(beta, dtype)) if affine_grad else (gamma.to(dtype).cuda(), beta.to(dtype).cuda())
    s, y = ops.add_layer_norm_pair(xd, rd, gd, bd, 1e-5)
    loss = (y * w1.to(dtype).cuda()).sum() + ((s * w2.to(dtype).cuda()).sum() if use_sum else 0)
    loss.backward()
    xr, rr, gr, br = (t.detach().float().cpu().requires_grad_() for t in (xd, rd, gd, bd))
    sr = (xr + rr).to(dtype).float() if dtype != torch.float32 else xr + rr
    sr = xr + rr + (sr - (xr + rr)).detach()          # the rounded sum, gradient of the plain sum
    yr = F.layer_norm(sr, (cols,), gr, br, 1e-5)
    lr = (yr * w1.to(dtype).float()).sum() + ((sr * w2.to(dtype).float()).sum() if use_sum else 0)
    lr.backward()
    t = tol(dtype)
    assert torch.equal(s, (xd + rd).detach())
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert torch.equal(xd.grad, rd.grad)
    if affine_grad:
        assert_close(gd.grad.float(), gr.grad, t, "dgamma")
        assert_close(bd.grad.float(), br.grad, t, "dbeta")


@pytest.mark.parametrize("rows,cols", [(37, 64), (640, 2048), (9, 1000)])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_res,affine_grad", [(True, True), (True, False), (False, True)])
def test_layer_norm_fanout(rows, cols, dtype, use_res, affine_grad):
    """(r, y) = (x, LN(x)) with both outputs feeding the loss (a pre-LN block: reference :316-320): dx = LN'(dy) + dr from ONE
    backward kernel, against torch's layer_norm + autograd's own accumulation in fp32."""
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(rows * 5 + cols)
    x = torch.randn(rows, cols, generator=g) * 1.5
    gamma, beta = torch.randn(cols, generator=g) * 0.2 + 1, torch.randn(cols, generator=g) * 0.1
    w1, w2 = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    xd = dev(x, dtype)
    gd, bd = (dev(gamma, dtype), dev(beta, dtype)) if affine_grad else (gamma.to(dtype).cuda(), beta.to(dtype).cuda())
    h = xd * 1.0                                      # a non-leaf input, as in the decoder loop
    r, y = ops.layer_norm_fanout(h, gd, bd, 1e-5)
    loss = (y * w1.to(dtype).cuda()).sum() + ((r * w2.to(dtype).cuda()).sum() if use_res else 0)
    loss.backward()
    xr, gr, br = (t.detach().float().cpu().requires_grad_() for t in (xd, gd, bd))
    yr = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    lr = (yr * w1.to(dtype).float()).sum() + ((xr * w2.to(dtype).float()).sum() if use_res else 0)
    lr.backward()
    t = tol(dtype)
    assert torch.equal(r.detach(), xd.detach())
    assert torch.equal(y.detach(), ops.layer_norm(xd.detach(), gd.detach(), bd.detach(), 1e-5))      # the same forward kernel
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    if affine_grad:
        assert_close(gd.grad.float(), gr.grad, t, "dgamma")
        assert_close(bd.grad.float(), br.grad, t, "dbeta")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [5, 4096, 640 * 2048 + 3])
def test_scale_by_device_scalar(dtype, n):
    """mmgl_scale: y = x * (*scale) with the product in fp32 (the 1 / grad_accumulation_steps of reference run_generation.py:483)."""
    from mmgl_amd import _lib
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g).to(dtype).cuda()
    s = torch.tensor([1.0 / 3.0], dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    _lib.call("mmgl_scale", None, _lib.ptr(x), _lib.ptr(s), _lib.ptr(y), n, _lib.dtype_code(x), _lib.stream_ptr())
    assert torch.equal(y, (x.float() * s).to(dtype))
    _lib.call("mmgl_scale", None, _lib.ptr(x), _lib.ptr(s), _lib.ptr(x), n, _lib.dtype_code(x), _lib.stream_ptr())       # in place
    assert torch.equal(x, y)


@pytest.mark.parametrize("dtype", DTYPES)
def test_add_layer_norm_pair_dropout_matches_unfused(dtype):
    """With dropout the fused pair must reproduce gated_residual(seed) -> layer_norm exactly in forward (same counter
    hash, same rounding points) and within tolerance in backward."""
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, cols, p, seed = 300, 2048, 0.3, 987654321
    x, r = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    gamma, beta = (torch.randn(cols, generator=g) * 0.2 + 1).to(dtype).cuda(), (torch.randn(cols, generator=g) * 0.1).to(dtype).cuda()
    w1, w2 = torch.randn(rows, cols, generator=g).to(dtype).cuda(), torch.randn(rows, cols, generator=g).to(dtype).cuda()
    xa, ra = dev(x, dtype), dev(r, dtype)
    s, y = ops.add_layer_norm_pair(xa, ra, gamma, beta, 1e-5, p, True, seed=seed)
    ((y * w1).sum() + (s * w2).sum()).backward()
    xb, rb = dev(x, dtype), dev(r, dtype)
    s2 = ops.gated_residual(rb, xb, None, p, True, seed=seed)
    y2 = ops.layer_norm(s2, gamma, beta, 1e-5)
    ((y2 * w1).sum() + (s2 * w2).sum()).backward()
    assert torch.equal(s, s2) and torch.equal(y, y2)
    dropped = (s.detach() == ra.detach()).float().mean().item()
    assert abs(dropped - p) < 0.02
    t = tol(dtype)
    assert_close(xa.grad.float(), xb.grad.float(), t, "dx")
    assert_close(ra.grad.float(), rb.grad.float(), t, "dres")
    if dtype == torch.float32:                       # dropped elements pass the residual through unchanged (in bf16 a small
        drop = s.detach() == ra.detach()             # kept x can round away too, so the forward cannot identify them)
        assert bool((xa.grad[drop] == 0).all()) and bool((xb.grad[drop] == 0).all())


@pytest.mark.parametrize("dtype", DTYPES)
def test_frozen_linear_relu(dtype):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(5)
    x = dev(torch.randn(3, 50, 64, generator=g), dtype)
    W = (torch.randn(96, 64, generator=g) * 0.2).to(dtype).cuda()
    b = (torch.randn(96, generator=g) * 0.2).to(dtype).cuda()
    w = torch.randn(3, 50, 96, generator=g).to(dtype).cuda()
    y = ops.frozen_linear_relu(x, W, b)
    (y * w).sum().backward()
    xr = x.detach().float().cpu().requires_grad_()
    yr = F.relu(F.linear(xr, W.float().cpu(), b.float().cpu()))
    keep = (y.detach().float().cpu() > 0) == (yr > 0)                    # rounding may flip a sign at pre-activation ~ 0
    (yr * w.float().cpu() * keep).sum().backward()
    t = tol(dtype, bf=3e-2)
    assert_close(y.float(), yr, t, "y")
    assert_close(x.grad.float(), xr.grad, t, "dx")
    with pytest.raises(ValueError):
        ops.frozen_linear_relu(x, W.clone().requires_grad_(), b)
    x2 = x.detach().clone().requires_grad_()              # plain frozen linear: dgrad against the cached W^T
    (ops.frozen_linear(x2, W, b) * w).sum().backward()
    xr2 = x.detach().float().cpu().requires_grad_()
    (F.linear(xr2, W.float().cpu(), b.float().cpu()) * w.float().cpu()).sum().backward()
    assert_close(x2.grad.float(), xr2.grad, t, "dx plain")


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_frozen_affine(dtype):
    from mmgl_amd import ops
    x = dev(torch.randn(50, 256), dtype)
    gamma = torch.randn(256).to(dtype).cuda()
    beta = torch.randn(256).to(dtype).cuda()
    y = ops.layer_norm(x, gamma, beta)
    y.sum().backward()
    xr = x.detach().float().cpu().requires_grad_()
    F.layer_norm(xr, (256,), gamma.float().cpu(), beta.float().cpu()).sum().backward()
    assert_close(x.grad.float(), xr.grad, tol(dtype, bf=3e-2), "dx")


@pytest.mark.parametrize("rows,cols", [(33, 128), (512, 4096)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_rmsnorm(rows, cols, dtype):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(rows, cols, generator=g)
    gamma = torch.randn(cols, generator=g) * 0.2 + 1
    w = torch.randn(rows, cols, generator=g)
    xd, gd = dev(x, dtype), dev(gamma, dtype)
    y = ops.rms_norm(xd, gd, 1e-6)
    (y * w.to(dtype).cuda()).sum().backward()
    xr, gr = (t.detach().float().cpu().requires_grad_() for t in (xd, gd))
    yr = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6) * gr
    (yr * w.to(dtype).float()).sum().backward()
    t = tol(dtype)
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert_close(gd.grad.float(), gr.grad, t, "dgamma")


@pytest.mark.parametrize("rows,cols", [(33, 128), (512, 4096)])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("use_sum,gamma_grad", [(True, True), (True, False), (False, False)])
def test_add_rms_norm_pair(rows, cols, dtype, use_sum, gamma_grad):
    """(s, y) = (x + r, RMSNorm(x + r)), both outputs feeding the loss: dx = dr = RMSNorm'(dy) + ds from one backward kernel."""
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    x, r = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g) * 2
    gamma = torch.randn(cols, generator=g) * 0.2 + 1
    w1, w2 = torch.randn(rows, cols, generator=g), torch.randn(rows, cols, generator=g)
    xd, rd = dev(x, dtype), dev(r, dtype)
    gd = dev(gamma, dtype) if gamma_grad else gamma.to(dtype).cuda()
    s, y = ops.add_rms_norm_pair(xd, rd, gd, 1e-6)
    loss = (y * w1.to(dtype).cuda()).sum() + ((s * w2.to(dtype).cuda()).sum() if use_sum else 0)
    loss.backward()
    xr, rr, gr = (t_.detach().float().cpu().requires_grad_() for t_ in (xd, rd, gd))
    sr = xr + rr
    sr = sr + ((xd + rd).detach().float().cpu() - sr).detach()          # the rounded sum, gradient of the plain sum
    yr = sr * torch.rsqrt(sr.pow(2).mean(-1, keepdim=True) + 1e-6) * gr
    lr = (yr * w1.to(dtype).float()).sum() + ((sr * w2.to(dtype).float()).sum() if use_sum else 0)
    lr.backward()
    t = tol(dtype)
    assert torch.equal(s, (xd + rd).detach())
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert torch.equal(xd.grad, rd.grad)
    if gamma_grad:
        assert_close(gd.grad.float(), gr.grad, t, "dgamma")


@pytest.mark.parametrize("n", [(3, 7, 64), (4, 640, 2048), (1, 5, 13)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gated_residual_eval(n, dtype):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(3)
    res, x, w = (torch.randn(*n, generator=g) for _ in range(3))
    gate = torch.tensor(0.7, requires_grad=True)
    rd, xd = dev(res, dtype), dev(x, dtype)
    gd = gate.detach().cuda().requires_grad_()
    y = ops.gated_residual(rd, xd, gd)
    (y * w.to(dtype).cuda()).sum().backward()
    rr, xr = (t.detach().float().cpu().requires_grad_() for t in (rd, xd))
    yr = rr + torch.tanh(gate) * xr
    (yr * w.to(dtype).float()).sum().backward()
    t = tol(dtype)
    assert_close(y.float(), yr, t, "y")
    assert_close(rd.grad.float(), rr.grad, t, "dres")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert_close(gd.grad.float().cpu(), gate.grad, tol(dtype, 1e-4, 2e-2), "dgate")
    # ungated form
    y2 = ops.gated_residual(rd, xd, None)
    assert_close(y2.float(), (rr + xr).detach(), t, "ungated")


def test_gated_residual_dropout_consistency():
    """Training-mode dropout: mask density ~ 1-p, scaling 1/(1-p), and backward regenerates the SAME mask."""
    from mmgl_amd import ops
    n, p = 1 << 20, 0.1
    res = torch.zeros(n, device="cuda")
    x = torch.ones(n, device="cuda", requires_grad=True)
    gate = torch.tensor(10.0, device="cuda")     # tanh ~ 1
    y = ops.gated_residual(res, x, gate, p_drop=p, training=True, seed=1234)
    kept = (y > 0).float().mean().item()
    assert abs(kept - (1 - p)) < 5e-3
    assert_close(y[y > 0].mean(), torch.tensor(1 / (1 - p)), 1e-3, "scale")
    y.sum().backward()
    assert torch.equal((x.grad > 0), (y > 0))
    y2 = ops.gated_residual(res, x.detach(), gate, p_drop=p, training=True, seed=1234)
    assert torch.equal(y, y2)
    y3 = ops.gated_residual(res, x.detach(), gate, p_drop=p, training=True, seed=99)
    assert not torch.equal(y, y3)


GEMM_CASES = [  # M, N, K
    (16, 64, 64), (37, 128, 72), (2560, 2048, 2048), (256, 2048, 2048), (640, 3072, 768), (44, 8192, 768), (300, 200, 136),
]


@pytest.mark.parametrize("M,N,K", GEMM_CASES)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ["none", "relu"])
def test_linear(M, N, K, dtype, act):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.3
    w = torch.randn(M, N, generator=g)
    scale = 0.125 if act == "none" else 1.0
    xd, Wd, bd = dev(x, dtype), dev(W, dtype), dev(b, dtype)
    if act == "relu":
        # keep the test away from the ReLU kink: outputs whose pre-activation is ~0 may legitimately land on either
        # side depending on summation order, so they get zero weight in the scalar that is differentiated
        pre = F.linear(xd.detach().float().cpu(), Wd.detach().float().cpu(), bd.detach().float().cpu())
        w = w * (pre.abs() > (1e-3 if dtype == torch.float32 else 5e-2))
    y = ops.linear(xd, Wd, bd, act=act, out_scale=scale)
    (y * w.to(dtype).cuda()).sum().backward()
    xr, Wr, br = (t.detach().float().cpu().requires_grad_() for t in (xd, Wd, bd))
    yr = F.linear(xr, Wr, br) * scale
    if act == "relu":
        yr = F.relu(yr)
    (yr * w.to(dtype).float()).sum().backward()
    t = tol(dtype, 1e-4, 2e-2)
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert_close(Wd.grad.float(), Wr.grad, t, "dW")
    assert_close(bd.grad.float(), br.grad, t, "db")


@pytest.mark.parametrize("act", ["none", "relu"])
@pytest.mark.parametrize("M,N,K", [(8192 - 64, 4096, 4096),      # big forward (ragged M), big dgrad, big wgrad (256 tiles)
                                   (8192, 2048, 2048),           # 64-tile weight gradient: split-K x4 + fp32 reduce
                                   (8192, 6504, 2048)])          # trainable lm_head-like: N no multiple of 128 / 256 -> zero-padded
                                                                 # contraction in dgrad, ragged last tile row in the weight gradient
def test_linear_big_tile_kernels(act, M, N, K):
    """Shapes large enough for the 256x256 kernels (>= 512 block tiles in forward AND in dgrad), ragged in M and N, checked
    against a CUDA fp32 matmul of the same bf16 inputs (the CPU reference of the small cases would take minutes here)."""
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, K, generator=g).bfloat16().cuda().requires_grad_()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).bfloat16().cuda().requires_grad_()
    b = (torch.randn(N, generator=g) * 0.3).bfloat16().cuda().requires_grad_()
    w = torch.randn(M, N, generator=g).bfloat16().cuda()
    xr, Wr, br = (t.detach().float().requires_grad_() for t in (x, W, b))
    yr = F.linear(xr, Wr, br)
    if act == "relu":
        w = w * (yr.detach().abs() > 5e-2)
        yr = F.relu(yr)
    (yr * w.float()).sum().backward()
    y = ops.linear(x, W, b, act=act)
    (y * w).sum().backward()
    assert_close(y.float().cpu(), yr.detach().cpu(), 2e-2, "y")
    assert_close(x.grad.float().cpu(), xr.grad.cpu(), 2e-2, "dx")
    assert_close(W.grad.float().cpu(), Wr.grad.cpu(), 2e-2, "dW")
    assert_close(b.grad.float().cpu(), br.grad.cpu(), 2e-2, "db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,F_", [(300, 64, 136), (8192, 2048, 4096)])
def test_linear_relu_pair_with_folded_mask(dtype, M, K, F_):
    """fc1+ReLU -> fc2 with fc1's ReLU backward folded into fc2's dgrad (bwd_premasked / mask_dx) must give the same
    gradients as the plain pair (the large case runs the 256x256 kernels' mask epilogue, the small one the in-place pass)."""
    from mmgl_amd import ops
    if dtype == torch.float32 and M > 1000:
        pytest.skip("large case: bf16 kernels only")
    g = torch.Generator().manual_seed(M + F_)
    x = (torch.randn(M, K, generator=g)).to(dtype).cuda()
    W1 = (torch.randn(F_, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b1 = (torch.randn(F_, generator=g) * 0.3).to(dtype).cuda()
    W2 = (torch.randn(K, F_, generator=g) * F_ ** -0.5).to(dtype).cuda()
    b2 = (torch.randn(K, generator=g) * 0.3).to(dtype).cuda()
    w = torch.randn(M, K, generator=g).to(dtype).cuda()
    grads = []
    for folded in (False, True):
        ps = [t.detach().clone().requires_grad_() for t in (x, W1, b1, W2, b2)]
        h = ops.linear(ps[0], ps[1], ps[2], act="relu", bwd_premasked=folded)
        y = ops.linear(h, ps[3], ps[4], mask_dx=folded)
        (y * w).sum().backward()
        grads.append([y.detach()] + [p.grad for p in ps])
    t = 1e-5 if dtype == torch.float32 else 1e-2
    for a, b, name in zip(grads[0], grads[1], ["y", "dx", "dW1", "db1", "dW2", "db2"]):
        assert_close(b.float(), a.float(), t, name)


def test_linear_a_is_identity_asymmetric():
    """Transpose-detecting check: W = asymmetric matrix, x = identity."""
    from mmgl_amd import ops
    N = K = 128
    W = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) / 251.0
    x = torch.eye(K)
    y = ops.linear(x.cuda(), W.cuda())
    assert_close(y, W.t(), 1e-6, "identity")


# (200, 256, 192): small ragged case; (2560, 2048, 2048) = BASELINE config 4 at the reference's batch 4 (OPT-1.3B q/v_proj, T = 640);
# (45056, 2048, 2048) = config 4 at the bench batch 64 (T = 640 + 64 concatenated neighbor tokens).  r = 16 as in config 4.
@pytest.mark.parametrize("M,N,K", [(200, 256, 192), (2560, 2048, 2048), (45056, 2048, 2048)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_lora_linear(dtype, M, N, K):
    from mmgl_amd import ops
    if dtype == torch.float32 and M > 10000:
        pytest.skip("fp32 parity path is exercised at the two smaller shapes")
    r, s = 16, 0.5
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    A = torch.randn(r, K, generator=g) * K ** -0.5
    Bm = torch.randn(N, r, generator=g) * 0.3
    w = torch.randn(M, N, generator=g) * (200.0 / M) ** 0.5          # keep dA / dB (sums over M rows) O(1)
    xd, Ad, Bd = dev(x, dtype), dev(A, dtype), dev(Bm, dtype)
    Wd, bd = W.to(dtype).cuda(), b.to(dtype).cuda()
    os_ = 0.125 if M % 2 == 0 else 1.0                                # the attention scaling of an adapted q_proj, in the epilogues
    y = ops.lora_linear(xd, Wd, bd, Ad, Bd, s, os_)
    (y * w.to(dtype).cuda()).sum().backward()
    xr, Ar, Br = (t.detach().float().requires_grad_() for t in (xd, Ad, Bd))       # fp32 torch reference (on the GPU: size)
    yr = (F.linear(xr, Wd.float(), bd.float()) + s * (xr @ Ar.t()) @ Br.t()) * os_
    (yr * w.to(dtype).float().cuda()).sum().backward()
    t = tol(dtype, 1e-4, 3e-2)
    assert_close(y.float(), yr, t, "y")
    assert_close(xd.grad.float(), xr.grad, t, "dx")
    assert_close(Ad.grad.float(), Ar.grad, t, "dA")
    assert_close(Bd.grad.float(), Br.grad, t, "dB")


def test_lora_qkv_fused_node_vs_fp32_definition():
    """q | k | v of a layer with LoRA on q_proj and v_proj as one node (ops.lora_qkv: fused base GEMM, low-rank updates accumulated in
    place, one dgrad GEMM) against the fp32 torch definition of the three projections, config-4 layer dims, 8192 rows."""
    from mmgl_amd import ops
    B, T, d, r, s, qs = 8, 1024, 2048, 16, 2.0, 0.125
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, T, d, generator=g)
    Ws = [torch.randn(d, d, generator=g) * d ** -0.5 for _ in range(3)]
    bs = [torch.randn(d, generator=g) * 0.1 for _ in range(3)]
    Aq, Av = (torch.randn(r, d, generator=g) * d ** -0.5 for _ in range(2))
    Bq, Bv = (torch.randn(d, r, generator=g) * 0.3 for _ in range(2))
    wgt = torch.randn(B, T, 3 * d, generator=g) * (200.0 / (B * T)) ** 0.5
    dt = torch.bfloat16
    xd, Aqd, Bqd, Avd, Bvd = (dev(t_, dt) for t_ in (x, Aq, Bq, Av, Bv))
    Wd = [w_.to(dt).cuda() for w_ in Ws]
    bd = [b_.to(dt).cuda() for b_ in bs]
    w_qkv = torch.cat([Wd[0].float() * qs, Wd[1].float(), Wd[2].float()], 0).to(dt).contiguous()
    b_qkv = torch.cat([bd[0].float() * qs, bd[1].float(), bd[2].float()], 0).to(dt).contiguous()
    assert ops.lora_qkv_supported(xd, w_qkv, r)
    qkv = ops.lora_qkv(xd, w_qkv, b_qkv, Aqd, Bqd, Avd, Bvd, s, qs)
    (qkv * wgt.to(dt).cuda()).sum().backward()
    xr, Aqr, Bqr, Avr, Bvr = (t_.detach().float().requires_grad_() for t_ in (xd, Aqd, Bqd, Avd, Bvd))
    q = (F.linear(xr, Wd[0].float(), bd[0].float()) + s * (xr @ Aqr.t()) @ Bqr.t()) * qs
    k = F.linear(xr, Wd[1].float(), bd[1].float())
    v = F.linear(xr, Wd[2].float(), bd[2].float()) + s * (xr @ Avr.t()) @ Bvr.t()
    ref = torch.cat([q, k, v], -1)
    (ref * wgt.to(dt).float().cuda()).sum().backward()
    assert_close(qkv.float(), ref, 3e-2, "qkv")
    assert_close(xd.grad.float(), xr.grad, 3e-2, "dx")
    for name, a_, b_ in (("dAq", Aqd, Aqr), ("dBq", Bqd, Bqr), ("dAv", Avd, Avr), ("dBv", Bvd, Bvr)):
        assert_close(a_.grad.float(), b_.grad, 3e-2, name)


@pytest.mark.parametrize("dtype", DTYPES)
def test_neighbor_interleave_vs_oracle(dtype):
    from mmgl_amd import ops
    from oracle import wrapper_ref
    fx = Fixture("g1_wrapper_all.npz")
    B, Nt, Ni, n, d = 2, 3, 2, 2, 64
    g = torch.Generator().manual_seed(2)
    te = torch.randn(B, Nt, n, d, generator=g)
    ve = torch.randn(B, Ni, n, d, generator=g)
    w = torch.randn(B, (Nt + Ni) * n, d, generator=g)
    b = fx.inp
    ted, ved = dev(te, dtype), dev(ve, dtype)
    out, valid = ops.neighbor_interleave(ted, ved, b["text_locations"].cuda(), b["image_locations"].cuda(),
                                         b["neighbor_pos_ids"].cuda(), b["neighbor_images_pos_ids"].cuda())
    (out * w.to(dtype).cuda()).sum().backward()
    ro, rv = wrapper_ref.interleave_neighbors(ted.detach().float().cpu(), ved.detach().float().cpu(), b["neighbor_pos_ids"],
                                              b["neighbor_images_pos_ids"], b["text_locations"], b["image_locations"])
    assert torch.equal(out.float().cpu(), ro)
    assert torch.equal(valid.bool().cpu(), rv)
    # gradient = gather of w
    wv = w.to(dtype).float().view(B, Nt + Ni, n, d)
    for bi in range(B):
        for j in range(Nt):
            assert torch.equal(ted.grad[bi, j].float().cpu(), wv[bi, b["text_locations"][bi, j]])
        for j in range(Ni):
            assert torch.equal(ved.grad[bi, j].float().cpu(), wv[bi, b["image_locations"][bi, j]])


@pytest.mark.parametrize("rows,V", [(46, 128), (1278, 50272), (7, 32000)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_cross_entropy(rows, V, dtype):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(rows)
    logits = torch.randn(rows, V, generator=g) * 3
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[::5] = -100
    ld = dev(logits, dtype)
    loss = ops.cross_entropy(ld, labels.cuda())
    (loss * 1.7).backward()
    lr = ld.detach().float().cpu().requires_grad_()
    lossr = F.cross_entropy(lr, labels)
    (lossr * 1.7).backward()
    assert_close(loss.cpu(), lossr, 1e-5 if dtype == torch.float32 else 1e-4, "loss")
    assert_close(ld.grad.float(), lr.grad, tol(dtype, 1e-4, 1e-2), "dlogits")


def test_position_ids():
    from mmgl_amd import ops
    from oracle import lm_ref
    g = torch.Generator().manual_seed(0)
    m = (torch.rand(5, 700, generator=g) > 0.3).long()
    m[0] = 1
    m[1] = 0
    assert torch.equal(ops.position_ids(m.cuda()).cpu(), lm_ref.learned_position_ids(m))


@pytest.mark.parametrize("dtype", DTYPES)
def test_adamw_matches_torch(dtype):
    from mmgl_amd import ops
    n = 10007
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n, generator=g)
    pref = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([pref], lr=1e-2, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-8)
    p = p0.to(dtype).cuda()
    master = p0.clone().cuda() if dtype == torch.bfloat16 else None
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        pref.grad = gr.to(dtype).float()
        opt.step()
        ops.adamw_step_(p, master, gr.to(dtype).cuda(), m, v, 1e-2, 0.9, 0.95, 1e-8, 0.01, step)
    assert_close((master if master is not None else p).float().cpu(), pref.detach(), 1e-5, "adamw")


def test_stream_ptr_is_torchs_current_stream():
    """_lib.stream_ptr() (the hipStream_t every C-ABI call is launched on) reads torch's current stream through torch._C directly: it
    must follow `with torch.cuda.stream(...)` exactly as torch.cuda.current_stream() does."""
    from mmgl_amd import _lib
    assert _lib.stream_ptr().value == torch.cuda.current_stream().cuda_stream or (_lib.stream_ptr().value is None and torch.cuda.current_stream().cuda_stream == 0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert _lib.stream_ptr().value == side.cuda_stream
    assert (_lib.stream_ptr().value or 0) == torch.cuda.current_stream().cuda_stream
