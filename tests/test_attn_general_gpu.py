"""The general attention core (mmgl_attn_general_*, csrc/attn_general.hip): layer_head_mask, output_attentions and attention-probability
dropout of MPTAttention.forward (reference model/modelling_cross_attention.py:237-256) on the HIP path, against the CPU oracle -- which
tests/test_oracle_golden.py pins to outputs of the reference itself for exactly these options (tests/golden/g10_attention_options_*).
The dropout mask is a counter hash on the HIP side and torch's RNG stream in the reference: mmgl_attn_dropout_mask exports the HIP
mask and the oracle applies THAT mask, so both sides drop the same probabilities."""
import pytest
import torch

from helpers import Fixture, assert_close
from oracle import lm_ref

pytestmark = pytest.mark.gpu


def _mk(B, H, T, S, D, causal, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, T, H * D, generator=g) * 0.4
    k = torch.randn(B, S, H * D, generator=g)
    v = torch.randn(B, S, H * D, generator=g)
    valid = torch.rand(B, S, generator=g) > 0.3
    valid[:, 0] = True
    if not causal and B > 1:
        valid[1, :] = False                    # a sample without any valid key: uniform over its S keys, halved dQ / dK
    w = torch.randn(B, T, H * D, generator=g)
    return q, k, v, valid, w


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [
    # B, H, T, S, D, causal, p_drop, head mask, probs
    (3, 4, 40, 12, 16, False, 0.0, True, True),
    (3, 4, 40, 12, 16, False, 0.25, True, True),
    (2, 2, 70, 130, 64, False, 0.1, False, False),
    (1, 2, 33, 64, 128, False, 0.5, True, False),
    (2, 4, 100, 100, 64, True, 0.1, True, True),
    (1, 2, 200, 200, 128, True, 0.0, False, True),
    (2, 3, 65, 65, 32, True, 0.3, True, False),
])
def test_attn_general_matches_oracle(shape, dtype):
    from mmgl_amd import ops
    B, H, T, S, D, causal, pd, use_hm, want_probs = shape
    q, k, v, valid, w = _mk(B, H, T, S, D, causal, seed=B * 1000 + T)
    hm = torch.tensor([1.0, 0.0, 0.5, 2.0, 1.5, 0.25][:H]) if use_hm else None
    seed = 987654321 + T
    qd, kd, vd = (x.to(dtype).cuda().requires_grad_() for x in (q, k, v))
    out, probs = ops.attn_general(qd, kd, vd, valid.cuda(), H, causal=causal, head_mask=None if hm is None else hm.cuda(), p_drop=pd,
                                  training=True, seed=seed, output_attentions=want_probs)
    (out * w.to(dtype).cuda()).sum().backward()
    # the oracle on the same (possibly bf16-rounded) inputs, in fp32, with the HIP side's keep mask
    keep = ops.attn_dropout_mask(B, H, T, S, pd, seed, "cuda").cpu().bool() if pd > 0 else None
    qr, kr, vr = (x.detach().float().cpu().requires_grad_() for x in (qd, kd, vd))
    m4 = lm_ref.decoder_self_mask(valid, torch.float32) if causal else lm_ref.expand_mask(valid, torch.float32, T)
    ro, rw = lm_ref.attention_core(qr, kr, vr, m4, H, head_mask=hm, keep=keep, p_drop=pd, return_probs=True)
    (ro * w.to(dtype).float()).sum().backward()
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(out).all()
    assert_close(out.float(), ro, tol, "out")
    if want_probs:
        assert probs.shape == (B, H, T, S) and not probs.requires_grad
        assert_close(probs.float(), rw, tol, "attention weights")
    else:
        assert probs is None
    assert_close(qd.grad.float(), qr.grad, tol, "dq")
    assert_close(kd.grad.float(), kr.grad, tol, "dk")
    assert_close(vd.grad.float(), vr.grad, tol, "dv")


def test_left_padded_causal_rows_without_an_allowed_key():
    """MPTAttention called directly with LEFT padding (the decoder itself asserts key 0 valid): the first rows of a sample see only padded
    keys -> every score is finfo.min after the clamp (uniform probabilities over all S keys), and torch.max splits the gradient at the
    tie 0.5 / 0.5 -- except on keys masked twice (future AND padded: finfo.min + finfo.min = -inf, the clamp constant wins outright).
    Reference model/modelling_cross_attention.py:51-79, 226-228; the oracle's torch.maximum reproduces both cases through autograd."""
    from mmgl_amd import ops
    B, H, T, D = 2, 2, 70, 16
    g = torch.Generator().manual_seed(77)
    q, k, v, w = (torch.randn(B, T, H * D, generator=g) * s for s in (0.4, 1.0, 1.0, 1.0))
    valid = torch.ones(B, T, dtype=torch.bool)
    valid[0, :5] = False                       # rows 0..4 of sample 0 have no allowed key
    valid[0, 40:43] = False                    # padded keys in the future of those rows: doubly masked
    valid[1, 0] = False
    valid[1, 66:] = False
    qd, kd, vd = (x.cuda().requires_grad_() for x in (q, k, v))
    out, _ = ops.attn_general(qd, kd, vd, valid.cuda(), H, causal=True)
    (out * w.cuda()).sum().backward()
    qr, kr, vr = (x.clone().requires_grad_() for x in (q, k, v))
    ro = lm_ref.attention_core(qr, kr, vr, lm_ref.decoder_self_mask(valid, torch.float32), H)
    (ro * w).sum().backward()
    assert_close(out, ro, 1e-4, "out")
    assert_close(qd.grad, qr.grad, 1e-4, "dq")
    assert_close(kd.grad, kr.grad, 1e-4, "dk")
    assert_close(vd.grad, vr.grad, 1e-4, "dv")
    # the doubly-masked keys receive gradient only from rows that do not see them as a tie: compare them on their own
    assert_close(kd.grad[0, 40:43], kr.grad[0, 40:43], 1e-4, "dk of the doubly-masked keys")


def test_head_mask_with_requires_grad_is_refused():
    from mmgl_amd import ops
    q, k, v, valid, _ = _mk(1, 2, 8, 8, 16, False, 1)
    hm = torch.ones(2, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError, match="head_mask"):
        ops.attn_general(q.cuda(), k.cuda(), v.cuda(), valid.cuda(), 2, head_mask=hm)
    with torch.no_grad():
        ops.attn_general(q.cuda(), k.cuda(), v.cuda(), valid.cuda(), 2, head_mask=hm)


def test_attn_dropout_mask_statistics_and_determinism():
    from mmgl_amd import ops
    B, H, T, S, p = 4, 8, 256, 64, 0.1
    m1 = ops.attn_dropout_mask(B, H, T, S, p, 11, "cuda")
    m2 = ops.attn_dropout_mask(B, H, T, S, p, 11, "cuda")
    m3 = ops.attn_dropout_mask(B, H, T, S, p, 12, "cuda")
    assert torch.equal(m1, m2)                                   # a function of (seed, index) only
    n = m1.numel()
    keep = m1.float().mean().item()
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-4, keep
    agree = (m1 == m3).float().mean().item()                     # independent masks agree with probability p^2 + (1 - p)^2
    assert abs(agree - (p * p + (1 - p) ** 2)) < 5e-3, agree
    per_head = m1.float().mean(dim=(0, 2, 3))
    assert (per_head - (1 - p)).abs().max().item() < 5e-3        # no head / row structure
    assert ops.attn_dropout_mask(1, 1, 8, 8, 0.0, 5, "cuda").all()
    # p = 0 in training and any p in eval mode are the identity
    q, k, v, valid, _ = _mk(2, 2, 16, 8, 16, False, 3)
    q, k, v, valid = q.cuda(), k.cuda(), v.cuda(), valid.cuda()
    a, _ = ops.attn_general(q, k, v, valid, 2, p_drop=0.3, training=False)
    b, _ = ops.attn_general(q, k, v, valid, 2, p_drop=0.0, training=True)
    c = ops.xattn_core(q, k, v, valid, 2)
    assert torch.equal(a, b)
    assert_close(a, c, 1e-5, "general core vs fused core without options")


@pytest.mark.parametrize("name", ["cross", "self"])
def test_mpt_attention_options_vs_reference_golden(name):
    """The module-level route: MPTAttention.forward(layer_head_mask=, output_attentions=True) in eval mode against G10's parameters and
    the oracle (G10 itself was taken WITH dropout under a fixed mask, which the HIP hash cannot replay: the oracle, pinned by G10 on the
    CPU side, is the bridge), then in training mode with attention dropout against the oracle under the exported HIP mask."""
    from mmgl_amd import ops
    from mmgl_amd.model.modelling_cross_attention import MPTAttention, MPTConfig
    from helpers import mpt_args, tiny_opt_config
    fx = Fixture(f"g10_attention_options_{name}.npz")
    H, pd = fx.meta["H"], fx.meta["p_drop"]
    cfg = MPTConfig(mpt_args(), tiny_opt_config())
    attn = MPTAttention(cfg, cross_attention=(name == "cross"))
    attn.load_state_dict(fx.p)
    attn = attn.cuda()
    hidden = fx.inp["hidden"].cuda()
    hm = fx.inp["head_mask"]
    valid = fx.inp["valid"]
    T = hidden.shape[1]
    p = {k: v.clone() for k, v in fx.p.items()}
    if name == "cross":
        ne = fx.inp["neighbor_embeds"]
        kw = dict(neighbor_embeds=ne.cuda(), neighbor_attention_mask=valid.cuda())
        m4 = lm_ref.expand_mask(valid, torch.float32, T)
    else:
        ne = None
        kw = dict(attention_mask=valid.cuda().to(torch.uint8))
        m4 = lm_ref.decoder_self_mask(valid, torch.float32)
    attn.eval()
    out, w, _ = attn(hidden, layer_head_mask=hm.cuda(), output_attentions=True, **kw)
    ro, rw = lm_ref.attention(p, "", fx.inp["hidden"], m4, H, kv_source=ne, head_mask=hm, return_probs=True)
    assert_close(out, ro, 1e-3, "eval out")
    assert_close(w, rw, 1e-3, "eval attention weights")
    # training mode, attention dropout on: same seed on both calls -> same mask; the oracle gets that mask
    attn.train()
    attn.dropout = pd
    torch.manual_seed(77)
    out1, _, _ = attn(hidden, layer_head_mask=hm.cuda(), **kw)
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())            # the draw ops.attn_general makes
    B, S = hidden.shape[0], (ne.shape[1] if ne is not None else T)
    keep = ops.attn_dropout_mask(B, H, T, S, pd, seed, "cuda").cpu().bool()
    ro1 = lm_ref.attention(p, "", fx.inp["hidden"], m4, H, kv_source=ne, head_mask=hm, keep=keep, p_drop=pd)
    assert_close(out1, ro1, 1e-3, "train out under the exported mask")
    assert not torch.equal(out1, out)


@pytest.mark.parametrize("r", [16, 4])
def test_lora_dropout_branch_matches_torch(r):
    """lora_dropout > 0 (reference model/modelling_self_attention.py:80-87 -> peft: base(x) + B(A(dropout(x))) * scaling, training only):
    LoRALinear's unfused route -- HIP dropout kernel, frozen base GEMM, two skinny GEMMs -- against torch autograd on the same mask
    (the mask is read back from the HIP dropout kernel under the seed the module drew)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from mmgl_amd import ops
    from mmgl_amd.model.modelling_self_attention import LoRALinear
    torch.manual_seed(3)
    M, K, N, p, alpha = 96, 64, 128, 0.2, 8.0
    base = nn.Linear(K, N)
    lora = LoRALinear(base, r, alpha, p).cuda()
    with torch.no_grad():
        lora.lora_B.normal_(std=0.1)
    x = torch.randn(2, M // 2, K, device="cuda", requires_grad=True)
    w = torch.randn(2, M // 2, N, device="cuda")
    lora.eval()
    y_eval = lora(x)
    ref_eval = F.linear(x, base.weight, base.bias) + (alpha / r) * F.linear(F.linear(x, lora.lora_A), lora.lora_B)
    assert_close(y_eval, ref_eval, 1e-3, "eval: dropout is the identity")
    lora.train()
    torch.manual_seed(41)
    y = lora(x)
    (y * w).sum().backward()
    got = (x.grad.clone(), lora.lora_A.grad.clone(), lora.lora_B.grad.clone())
    torch.manual_seed(41)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    keep = ops.gated_residual(torch.zeros_like(x), torch.ones_like(x), None, p, True, seed=seed).detach() != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.03
    xr = x.detach().clone().requires_grad_()
    A = lora.lora_A.detach().clone().requires_grad_()
    Bm = lora.lora_B.detach().clone().requires_grad_()
    xd = xr * keep / (1 - p)
    ref = F.linear(xr, base.weight.detach(), base.bias.detach()) + (alpha / r) * F.linear(F.linear(xd, A), Bm)
    (ref * w).sum().backward()
    assert_close(y, ref, 1e-3, "train out")
    assert_close(got[0], xr.grad, 1e-3, "dx")
    assert_close(got[1], A.grad, 1e-3, "d lora_A")
    assert_close(got[2], Bm.grad, 1e-3, "d lora_B")
