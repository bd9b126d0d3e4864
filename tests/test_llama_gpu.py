"""-m gpu: Llama-convention variant (BASELINE.json configs[4] family).  No reference counterpart exists, so the pins are:
gates = 0  => logits == the frozen HF LlamaForCausalLM (fp32, CPU);  gates != 0 => the CPU oracle block (oracle/llama_ref.py)."""
import copy

import pytest
import torch

from helpers import assert_close, mpt_args, tiny_clip_vision_config, tiny_roberta_config

pytestmark = pytest.mark.gpu


def _tiny_llama():
    from transformers import LlamaConfig
    return LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=4, num_attention_heads=4,
                       num_key_value_heads=4, max_position_embeddings=256, pad_token_id=1, bos_token_id=2, eos_token_id=2,
                       attention_dropout=0.0)


def _batch(B=2, T=24, S=10, d=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 128, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    am[1, T - 6:] = 0
    ne = torch.randn(B, S, d, generator=g)
    valid = torch.ones(B, S, dtype=torch.bool)
    valid[0, 6:] = False
    valid[1, 1::3] = False
    return ids, am, ne, valid


def _build():
    from mmgl_amd.model.modelling_llama_cross_attention import LlamaNeighborLM
    torch.manual_seed(0)
    lm = LlamaNeighborLM(mpt_args(model_name_or_path="llama-tiny", neighbor_layer_wise=2), _tiny_llama())
    return lm


def test_llama_gates_zero_equals_hf_llama():
    lm = _build()
    hf = copy.deepcopy(lm.llama).float().eval()
    ids, am, ne, valid = _batch()
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=am).logits
        got = lm.cuda().eval()(input_ids=ids.cuda(), attention_mask=am.cuda(), labels=ids.cuda(), neighbor_embeds=ne.cuda(),
                               neighbor_attention_mask=valid.cuda()).logits
    assert_close(got, want, 1e-3, "gates=0 logits vs HF Llama")


def test_llama_gated_block_vs_oracle_fwd_bwd():
    from oracle import llama_ref
    lm = _build()
    with torch.no_grad():
        for i, layer in enumerate(lm.neighbor_layers):
            layer.gating1.fill_(0.5 + 0.1 * i)
            layer.gating2.fill_(-0.3 - 0.1 * i)
            layer.input_layernorm.add_(0.1 * torch.randn(64))
    hf = copy.deepcopy(lm.llama).float().eval()
    p = {k: v.detach().clone().float().requires_grad_() for k, v in lm.state_dict().items() if k.startswith("neighbor_layers.")}
    ids, am, ne, valid = _batch(seed=3)
    logits, loss = llama_ref.llama_neighbor_lm_forward(hf, p, 2, ids, am, ids, ne, valid)
    loss.backward()
    lm = lm.cuda().eval()
    out = lm(input_ids=ids.cuda(), attention_mask=am.cuda(), labels=ids.cuda(), neighbor_embeds=ne.cuda(), neighbor_attention_mask=valid.cuda())
    assert_close(out.logits, logits, 1e-3, "logits")
    assert_close(out.loss, loss, 1e-3, "loss")
    out.loss.backward()
    trainable = {k for k, q in lm.named_parameters() if q.requires_grad}
    assert trainable == {k for k in p}, "only the gated layers are trainable"
    for k, q in lm.named_parameters():
        if q.requires_grad:
            assert_close(q.grad, p[k].grad, 2e-3, f"d {k}")


def test_cross_attention_model_selects_llama_variant():
    from mmgl_amd.model import CrossAttentionModel
    w = CrossAttentionModel(mpt_args(model_name_or_path="llama-tiny", context="text_only", neighbor_layer_wise=2), None,
                            lm_config=_tiny_llama(), text_config=tiny_roberta_config(), visual_config=tiny_clip_vision_config())
    from mmgl_amd.model.modelling_llama_cross_attention import LlamaNeighborLM
    assert isinstance(w.lm, LlamaNeighborLM)
    w = w.cuda().bfloat16().train()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 128, (2, 24), generator=g).cuda()
    am = torch.ones(2, 24, dtype=torch.long).cuda()
    nids = torch.randint(3, 128, (2, 3, 12), generator=g).cuda()
    nam = torch.ones(2, 3, 12, dtype=torch.long).cuda()
    npos = torch.tensor([[1, 2, 0], [1, 0, 0]]).cuda()
    out = w(input_ids=ids, attention_mask=am, labels=ids, neighbor_input_ids=nids, neighbor_attention_mask=nam, neighbor_pos_ids=npos)
    out.loss.backward()
    assert torch.isfinite(out.loss)
    assert all(p.grad is not None for n, p in w.named_parameters() if p.requires_grad and "neighbor_layers" in n)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,T,H,D", [(2, 24, 4, 16), (1, 2176, 8, 128)])
def test_rope_qk_matches_transformers_rotate_half(B, T, H, D, dtype, tol):
    """mmgl_rope_inplace on the q / k thirds of a fused-QKV buffer == transformers' apply_rotary_pos_emb (rotate_half), forward
    and gradient; the v third is untouched."""
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(T)
    d = H * D
    qkv = torch.randn(B, T, 3 * d, generator=g)
    w = torch.randn(B, T, 3 * d, generator=g)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(T).float()[:, None] * inv[None]
    cos_sin = torch.stack([ang.cos(), ang.sin()], -1)
    x = qkv.to(dtype).cuda().requires_grad_()
    y = ops.rope_qk_(x * 1.0, cos_sin.cuda(), H)            # (x * 1.0: a fresh buffer, as the GEMM output is)
    (y.float() * w.cuda()).sum().backward()
    xr = x.detach().float().cpu().requires_grad_()
    q, k, v = (xr[..., i * d:(i + 1) * d].view(B, T, H, D).transpose(1, 2) for i in range(3))
    cos = torch.cat([ang.cos(), ang.cos()], -1)[None]
    sin = torch.cat([ang.sin(), ang.sin()], -1)[None]
    qe, ke = apply_rotary_pos_emb(q, k, cos, sin)
    yr = torch.cat([t.transpose(1, 2).reshape(B, T, d) for t in (qe, ke, v)], -1)
    (yr * w).sum().backward()
    assert_close(y.float(), yr, tol, "rope fwd")
    assert_close(x.grad.float(), xr.grad, tol, "rope bwd")
    assert torch.equal(y[..., 2 * d:].float().cpu(), x.detach()[..., 2 * d:].float().cpu())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("M,F", [(50, 128), (2176, 11008)])
def test_swiglu_fwd_bwd(M, F, dtype, tol):
    from mmgl_amd import ops
    g = torch.Generator().manual_seed(F)
    gu = torch.randn(M, 2 * F, generator=g)
    w = torch.randn(M, F, generator=g)
    x = gu.to(dtype).cuda().requires_grad_()
    y = ops.swiglu(x)
    (y.float() * w.cuda()).sum().backward()
    xr = x.detach().float().cpu().requires_grad_()
    yr = torch.nn.functional.silu(xr[:, :F]) * xr[:, F:]
    (yr * w).sum().backward()
    assert_close(y.float(), yr, tol, "swiglu fwd")
    assert_close(x.grad.float(), xr.grad, tol, "swiglu bwd")


def test_llama_config5_dims_one_layer_bf16():
    """One frozen layer + one gated block at Llama-2-7B's dims (d 4096, 32 heads x 128, ffn 11008), T = 2176, S = 128, bf16:
    the frozen path against HF's own LlamaDecoderLayer arithmetic in fp32 (gates 0), then gradients flow with gates != 0."""
    from transformers import LlamaConfig
    from mmgl_amd.model.modelling_llama_cross_attention import LlamaNeighborLM
    cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32,
                      num_key_value_heads=32, max_position_embeddings=4096, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    torch.manual_seed(0)
    lm = LlamaNeighborLM(mpt_args(model_name_or_path="llama-2-7b", neighbor_layer_wise=1), cfg)
    B, T, S = 1, 2176, 128
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 32000, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    am[0, 1500:2048] = 0
    ne = torch.randn(B, S, 4096, generator=g) * 0.5
    valid = torch.ones(B, S, dtype=torch.bool)
    valid[0, 100:] = False
    lm = lm.bfloat16().cuda().eval()
    with torch.no_grad():
        got = lm(input_ids=ids.cuda(), attention_mask=am.cuda(), neighbor_embeds=ne.cuda(), neighbor_attention_mask=valid.cuda()).logits
        hf = lm.llama.float()
        hf.model.rotary_emb.inv_freq.copy_(lm._inv_freq)     # the bf16 cast above rounded HF's frequency buffer too: undo that (see LlamaNeighborLM.__init__)
        want = hf(input_ids=ids.cuda(), attention_mask=am.cuda()).logits                          # HF forward, fp32, same (bf16-rounded) weights
    keep = am.bool().cuda()
    assert_close(got.float()[keep], want[keep], 3e-2, "config-5 dims: logits vs HF Llama (valid positions)")
    lm.llama.bfloat16()
    with torch.no_grad():
        lm.neighbor_layers[0].gating1.fill_(0.5)
        lm.neighbor_layers[0].gating2.fill_(0.3)
    out = lm(input_ids=ids.cuda(), attention_mask=am.cuda(), labels=ids.cuda(), neighbor_embeds=ne.cuda().bfloat16(), neighbor_attention_mask=valid.cuda())
    out.loss.backward()
    for n, p in lm.neighbor_layers.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, n
