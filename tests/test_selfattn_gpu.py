"""-m gpu parity: causal self-attention flash kernels (C ABI) vs the CPU oracle's additive-mask attention
(oracle/lm_ref.py: decoder_self_mask + attention_core, i.e. the reference's :51-79, :206-235 semantics)."""
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu

CASES = [  # B, H, T, D
    (2, 2, 24, 16),        # tiny (golden LM shape)
    (2, 3, 100, 32),       # ragged T
    (3, 4, 640, 64),       # OPT shape
    (1, 2, 1000, 128),     # Llama head dim, T not a tile multiple
    (1, 4, 2176, 128),     # BASELINE config 5: Llama-2-7B head dim at T = 2048 + 128
]


def _oracle(q, k, v, am, H, w):
    from oracle import lm_ref
    q, k, v = (t.detach().float().cpu().requires_grad_() for t in (q, k, v))
    mask = lm_ref.decoder_self_mask(am.cpu(), torch.float32)
    out = lm_ref.attention_core(q, k, v, mask, H)
    (out * w.float().cpu()).sum().backward()
    return out.detach(), q.grad, k.grad, v.grad


@pytest.mark.parametrize("B,H,T,D", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_selfattn_fwd_bwd_vs_oracle(B, H, T, D, dtype):
    from mmgl_amd import ops
    gen = torch.Generator().manual_seed(B * 1000 + T)
    d = H * D
    q = torch.randn(B, T, d, generator=gen) * (D ** -0.5) * 2
    k = torch.randn(B, T, d, generator=gen)
    v = torch.randn(B, T, d, generator=gen)
    w = torch.randn(B, T, d, generator=gen)
    am = torch.ones(B, T, dtype=torch.long)
    am[0, T // 2: T - T // 5] = 0                   # WikiWeb2M layout: prompt | pad | summary | pad
    am[0, T - T // 10:] = 0
    if B > 1:
        am[1, T // 3:] = 0
    qd, kd, vd = (t.to(dtype).cuda().requires_grad_() for t in (q, k, v))
    out = ops.selfattn_core(qd, kd, vd, am.cuda(), H)
    (out * w.to(dtype).cuda()).sum().backward()
    ro, rq, rk, rv = _oracle(qd, kd, vd, am, H, w.to(dtype))
    assert torch.isfinite(out).all()
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    assert_close(out.float(), ro, tol, "out")
    assert_close(qd.grad.float(), rq, tol, "dq")
    assert_close(kd.grad.float(), rk, tol, "dk")
    assert_close(vd.grad.float(), rv, tol, "dv")


def test_selfattn_causality_and_determinism():
    """Size-independent properties at OPT-1.3B shape: outputs at rows < t do not depend on tokens >= t; masked keys do
    not matter; repeated launches are bitwise identical (forward and backward)."""
    from mmgl_amd import ops
    B, H, T, D = 2, 32, 640, 64
    gen = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, T, H * D, generator=gen).cuda() * 0.3 for _ in range(3))
    am = torch.ones(B, T, dtype=torch.long)
    am[:, 400:512] = 0
    am = am.cuda()
    o1 = ops.selfattn_core(q, k, v, am, H)
    k2, v2, q2 = k.clone(), v.clone(), q.clone()
    k2[:, 300:] = 7.0
    v2[:, 300:] = -7.0
    q2[:, 300:] = 1.0
    o2 = ops.selfattn_core(q2, k2, v2, am, H)
    assert torch.equal(o1[:, :300], o2[:, :300])
    k3, v3 = k.clone(), v.clone()
    k3[:, 400:512] = 1e3
    v3[:, 400:512] = -1e3
    assert torch.equal(o1, ops.selfattn_core(q, k3, v3, am, H))
    qg = q.clone().requires_grad_()
    g1 = torch.autograd.grad(ops.selfattn_core(qg, k, v, am, H).sum(), qg)[0]
    g2 = torch.autograd.grad(ops.selfattn_core(qg, k, v, am, H).sum(), qg)[0]
    assert torch.equal(g1, g2)


@pytest.mark.parametrize("B,H,T,D", [(3, 4, 640, 64), (2, 2, 2176, 128)])
def test_selfattn_dead_key_tiles_are_never_touched(B, H, T, D):
    """Key tiles whose 64 keys are all padding leave the walk of the forward and dQ kernels, key blocks whose 128 keys are all padding
    leave the dK / dV launch (round 6; WikiWeb2M pads every prompt to max_input_length, reference wikiweb2m/data.py:320-321).  NaNs in
    the K / V rows of fully dead tiles must not reach any output (a tile that is fetched and multiplied by zero probabilities would
    turn them into NaN), outputs and gradients are bit-identical to the clean run, dK / dV of padded keys are exactly zero."""
    from mmgl_amd import ops
    gen = torch.Generator().manual_seed(B + T)
    d = H * D
    q, k, v, w = ((torch.randn(B, T, d, generator=gen) * s).bfloat16().cuda() for s in (0.3, 1.0, 1.0, 1.0))
    am = torch.ones(B, T, dtype=torch.long)
    am[0, 70:T - 128] = 0                     # prompt | pad ... | summary: mixed tile 64..127, dead tiles 128 .. T-129
    am[0, T - 40:] = 0
    am[1, 200:] = 0                           # everything after the prompt is padding
    dead = torch.zeros(B, T, dtype=torch.bool)
    dead[0, 128:T - 128] = True
    dead[1, 256:] = True
    am_d = am.cuda()

    def run(kk, vv):
        qq, kk, vv = (t.clone().requires_grad_() for t in (q, kk, vv))
        o = ops.selfattn_core(qq, kk, vv, am_d, H)
        return (o,) + torch.autograd.grad((o * w).sum(), (qq, kk, vv))

    clean = run(k, v)
    kp, vp = k.clone(), v.clone()
    kp[dead.cuda()] = float("nan")
    vp[dead.cuda()] = float("nan")
    poisoned = run(kp, vp)
    for name, a, b in zip(("out", "dq", "dk", "dv"), clean, poisoned):
        assert torch.isfinite(b).all(), f"{name}: a dead key tile was touched"
        assert torch.equal(a, b), name
    pad = ~am.bool().cuda()
    assert (clean[2][pad] == 0).all() and (clean[3][pad] == 0).all()


@pytest.mark.parametrize("B,H,T,D", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_selfattn_fused_qkv_is_bitwise_the_separate_path(B, H, T, D, dtype):
    """Q/K/V read in place from one [B,T,3d] buffer (row stride 3d) and dQ/dK/dV written into one: same kernels, same
    values -> bit-identical to the packed call, forward and backward."""
    from mmgl_amd import ops
    d = H * D
    gen = torch.Generator().manual_seed(B * 7 + T)
    qkv = (torch.randn(B, T, 3 * d, generator=gen) * 0.4).to(dtype).cuda().requires_grad_()
    am = torch.ones(B, T, dtype=torch.long)
    am[0, T // 2: T // 2 + 5] = 0
    am = am.cuda()
    w = torch.randn(B, T, d, generator=gen).to(dtype).cuda()
    out_f = ops.selfattn_core_fused(qkv, am, H)
    (g_f,) = torch.autograd.grad((out_f * w).sum(), qkv)
    q, k, v = (qkv.detach()[..., i * d:(i + 1) * d].contiguous().requires_grad_() for i in range(3))
    out_s = ops.selfattn_core(q, k, v, am, H)
    gq, gk, gv = torch.autograd.grad((out_s * w).sum(), (q, k, v))
    assert torch.equal(out_f, out_s)
    assert torch.equal(g_f, torch.cat([gq, gk, gv], -1))


PREFIX_CASES = [  # B, H, T, P, D
    (2, 2, 24, 4, 16),       # tiny
    (2, 3, 100, 20, 32),     # peft's 20 virtual tokens, ragged T
    (3, 4, 640, 20, 64),     # OPT shape: P = 20 is no multiple of a tile (three diagonal steps in the dK/dV kernel)
    (1, 2, 300, 64, 128),    # a whole key tile of prefix, head dim 128
]


@pytest.mark.parametrize("B,H,T,P,D", PREFIX_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_selfattn_prefix_fwd_bwd_vs_oracle(B, H, T, P, D, dtype):
    """Prefix tuning's attention: P learned keys / values in front of the layer's own, visible to every query; the T causal keys
    keep their padding mask.  Oracle: the reference's additive-mask attention core over the concatenated keys."""
    from mmgl_amd import ops
    from oracle import lm_ref
    gen = torch.Generator().manual_seed(B * 1000 + T + P)
    d = H * D
    q = torch.randn(B, T, d, generator=gen) * (D ** -0.5) * 2
    k = torch.randn(B, P + T, d, generator=gen)
    v = torch.randn(B, P + T, d, generator=gen)
    w = torch.randn(B, T, d, generator=gen)
    am = torch.ones(B, P + T, dtype=torch.long)
    am[0, P + T // 2: P + T - T // 5] = 0
    if B > 1:
        am[1, P + T // 3:] = 0
    qd, kd, vd = (t.to(dtype).cuda().requires_grad_() for t in (q, k, v))
    out = ops.selfattn_core_prefix(qd, kd, vd, am.cuda(), H, P)
    (out * w.to(dtype).cuda()).sum().backward()
    # additive mask [B, 1, T, P + T]: key s allowed for query t iff s <= t + P and am[b, s]
    s_idx, t_idx = torch.arange(P + T)[None, :], torch.arange(T)[:, None]
    allowed = (s_idx <= t_idx + P)[None] & am.bool()[:, None, :]
    mask = torch.where(allowed, 0.0, torch.finfo(torch.float32).min)[:, None]
    qr, kr, vr = (t.detach().float().cpu().requires_grad_() for t in (qd, kd, vd))
    ro = lm_ref.attention_core(qr, kr, vr, mask, H)
    (ro * w.to(dtype).float()).sum().backward()
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(out).all()
    assert_close(out.float(), ro.detach(), tol, "out")
    assert_close(qd.grad.float(), qr.grad, tol, "dq")
    assert_close(kd.grad.float(), kr.grad, tol, "dk")
    assert_close(vd.grad.float(), vr.grad, tol, "dv")
