"""Packed (padding-free) frozen-encoder forward vs the HF modules it replaces, and its three kernels vs torch fp32.

The HF modules here are the very objects the reference calls (`self.text_model(...)`, `self.visual_model(...)`,
reference model/modelling_cross_attention.py:992, 1018); random-init, no checkpoint needed.  Tolerances: fp32 2e-4
relative to the output scale (different summation order + online softmax), bf16 4e-2.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


def _ragged_lens(n, L, gen):
    lens = torch.randint(1, L + 1, (n,), generator=gen)
    lens[0] = L
    if n > 1:
        lens[1] = 1
    return lens


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("H,D", [(4, 16), (2, 32), (12, 64), (2, 128)])
def test_encoder_attention_vs_torch(dtype, tol, H, D):
    from mmgl_amd import ops
    gen = torch.Generator().manual_seed(H * 100 + D)
    n, L = 7, 150
    lens = _ragged_lens(n, L, gen)
    cu = torch.zeros(n + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    ntok = int(cu[-1])
    qkv = torch.randn(ntok, 3 * H * D, generator=gen).to(dtype).cuda()
    hd = H * D
    q, k, v = qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:]
    scale = 1.0 / math.sqrt(D)
    qs = (qkv.float() * 1).clone()
    qs[:, :hd] *= scale
    qs = qs.to(dtype)
    out = ops.encoder_attention(qs[:, :hd], qs[:, hd:2 * hd], qs[:, 2 * hd:], cu.cuda(), H, int(lens.max()))
    out1 = ops.encoder_attention(qs[:, :hd], qs[:, hd:2 * hd], qs[:, 2 * hd:], cu.cuda(), H, int(lens.max()), q_rows=1)
    for i in range(n):
        a, b = int(cu[i]), int(cu[i + 1])
        qi = qs[a:b, :hd].float().view(b - a, H, D).transpose(0, 1)
        ki = k[a:b].float().view(b - a, H, D).transpose(0, 1)
        vi = v[a:b].float().view(b - a, H, D).transpose(0, 1)
        ref = torch.softmax(qi @ ki.transpose(1, 2), -1) @ vi
        ref = ref.transpose(0, 1).reshape(b - a, hd)
        assert _rel(out[a:b], ref) < tol, (i, b - a)
        assert _rel(out1[a:a + 1], ref[:1]) < tol, (i, b - a)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("cols", [64, 768, 1024, 4096])
def test_add_layer_norm(dtype, tol, cols):
    from mmgl_amd import ops
    torch.manual_seed(cols)
    x, r = torch.randn(37, cols).to(dtype).cuda(), torch.randn(37, cols).to(dtype).cuda()
    g, b = torch.randn(cols).to(dtype).cuda(), torch.randn(cols).to(dtype).cuda()
    s, y = ops.add_layer_norm(x, r, g, b, 1e-5, return_sum=True)
    s_ref = x + r
    assert torch.equal(s, s_ref)
    y_ref = torch.nn.functional.layer_norm(s_ref.float(), (cols,), g.float(), b.float(), 1e-5)
    assert _rel(y, y_ref) < tol
    assert _rel(ops.add_layer_norm(x, r, g, b, 1e-5), y_ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("name", ["relu", "gelu", "quick_gelu", "gelu_new"])
def test_activation(dtype, tol, name):
    from transformers.activations import ACT2FN
    from mmgl_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(33, 1001) * 3).to(dtype).cuda()
    ref = ACT2FN[name](x.float())
    y = ops.activation_(x.clone(), name)
    assert _rel(y, ref) < tol


def _text_model(hidden, heads, layers, inter, vocab=120, L=48):
    from transformers import RobertaConfig, RobertaModel
    cfg = RobertaConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                        intermediate_size=inter, max_position_embeddings=L + 2, hidden_dropout_prob=0.1,
                        attention_probs_dropout_prob=0.1)
    torch.manual_seed(7)
    return RobertaModel(cfg).eval()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("hidden,heads,layers,inter", [(64, 4, 3, 128), (768, 12, 2, 3072)])
def test_packed_text_encoder_matches_hf(dtype, tol, hidden, heads, layers, inter):
    from mmgl_amd.model.encoders import PackedTextEncoder
    L = 48
    model = _text_model(hidden, heads, layers, inter, L=L).to(dtype).cuda()
    assert PackedTextEncoder.supports(model)
    gen = torch.Generator().manual_seed(3)
    n = 9
    lens = _ragged_lens(n, L, gen)
    ids = torch.randint(3, 120, (n, L), generator=gen)
    am = (torch.arange(L)[None] < lens[:, None]).long()
    ids = torch.where(am.bool(), ids, torch.ones_like(ids))          # pad id 1 where masked
    ids[:, 0] = 0
    ids, am = ids.cuda(), am.cuda()
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=am).last_hidden_state[:, 0]
    got = PackedTextEncoder(model).cls(ids, am)
    assert got is not None and got.shape == ref.shape
    assert _rel(got, ref) < tol
    # a sequence whose first token is masked is not packable: there is no fallback: it raises
    am2 = am.clone()
    am2[2, 0] = 0
    with pytest.raises(ValueError):
        PackedTextEncoder(model).cls(ids, am2)
    # host metadata (lengths known from the collate) gives the same result without the device->host copy
    lens_cpu = am.sum(1).to(torch.int32).cpu()
    got2 = PackedTextEncoder(model).cls(ids, am, (lens_cpu, True))
    assert torch.equal(got2, PackedTextEncoder(model).cls(ids, am))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("hidden,heads,layers,inter,img,patch", [(64, 4, 3, 128, 32, 8), (768, 12, 2, 3072, 224, 16)])
def test_packed_vision_encoder_matches_hf(dtype, tol, hidden, heads, layers, inter, img, patch):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from mmgl_amd.model.encoders import PackedVisionEncoder
    cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                           image_size=img, patch_size=patch)
    torch.manual_seed(11)
    model = CLIPVisionModel(cfg).eval().to(dtype).cuda()
    assert PackedVisionEncoder.supports(model)
    pv = torch.randn(5, 3, img, img, generator=torch.Generator().manual_seed(5)).to(dtype).cuda()
    with torch.no_grad():
        ref = model(pv).pooler_output
    got = PackedVisionEncoder(model).pooled(pv)
    assert got is not None and got.shape == ref.shape
    assert _rel(got, ref) < tol


def test_fused_weights_follow_parameter_updates():
    from mmgl_amd.model.encoders import PackedTextEncoder
    model = _text_model(64, 4, 2, 128).cuda()
    enc = PackedTextEncoder(model)
    ids = torch.randint(3, 120, (4, 48)).cuda()
    ids[:, 0] = 0
    am = torch.ones_like(ids)
    a = enc.cls(ids, am)
    with torch.no_grad():
        model.encoder.layer[0].attention.self.query.weight.normal_(0, 1.0)
        model.embeddings.position_embeddings.weight.normal_(0, 0.5)
    b = enc.cls(ids, am)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=am).last_hidden_state[:, 0]
    assert _rel(b, ref) < 2e-4 and _rel(a, ref) > 1e-3


def test_encoder_attention_sequences_are_independent():
    """Size-independent property at RoBERTa-base width: a sequence's output does not depend on what else is in the pack,
    nor on where in the pack it sits (bit-exact)."""
    from mmgl_amd import ops
    H, D = 12, 64
    hd = H * D
    gen = torch.Generator().manual_seed(21)
    lens_a = torch.tensor([512, 37, 300, 1, 64])
    lens_b = torch.tensor([5, 300, 512, 129])                   # sequence of length 300 is shared: a[2] == b[1]
    shared = torch.randn(300, 3 * hd, generator=gen).bfloat16()

    def pack(lens, slot):
        parts = [torch.randn(int(n), 3 * hd, generator=gen).bfloat16() for n in lens]
        parts[slot] = shared
        cu = torch.zeros(len(lens) + 1, dtype=torch.int32)
        cu[1:] = torch.cumsum(lens, 0)
        qkv = torch.cat(parts).cuda()
        out = ops.encoder_attention(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], cu.cuda(), H, int(lens.max()))
        return out[int(cu[slot]):int(cu[slot + 1])]

    assert torch.equal(pack(lens_a, 2), pack(lens_b, 1))


def test_packed_text_encoder_ignores_padding_width():
    """The packed pass never touches pad tokens: the same sequences padded to a wider L give bit-identical CLS states."""
    from mmgl_amd.model.encoders import PackedTextEncoder
    model = _text_model(768, 12, 2, 3072, L=96).bfloat16().cuda()
    gen = torch.Generator().manual_seed(4)
    n, L = 6, 48
    lens = _ragged_lens(n, L, gen)
    ids = torch.randint(3, 120, (n, L), generator=gen)
    am = (torch.arange(L)[None] < lens[:, None]).long()
    ids = torch.where(am.bool(), ids, torch.ones_like(ids))
    ids[:, 0] = 0
    wide_ids = torch.cat([ids, torch.ones(n, 40, dtype=ids.dtype)], 1)
    wide_am = torch.cat([am, torch.zeros(n, 40, dtype=am.dtype)], 1)
    enc = PackedTextEncoder(model)
    a = enc.cls(ids.cuda(), am.cuda())
    b = enc.cls(wide_ids.cuda(), wide_am.cuda())
    assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("hidden,heads,layers,inter,eos", [(64, 4, 3, 128, 2), (512, 8, 2, 2048, 119)])
def test_clip_text_encoder_matches_hf(dtype, tol, hidden, heads, layers, inter, eos):
    """CLIP's text tower (reference model/modelling_cross_attention.py:918-921 accepts a CLIPTextModel as text_model): causal
    pre-LN encoder, pooled output = final_layer_norm(h)[EOS position], on the HIP kernels (no HF / SDPA / library-GEMM forward).
    eos_token_id 2 = the legacy argmax(ids) rule, otherwise the first position holding the EOS id (CLIP ViT-B/32 dims: 512 x 8)."""
    from transformers import CLIPTextConfig, CLIPTextModel
    from mmgl_amd.model.encoders import ClipTextEncoder
    L, vocab = 24, 120
    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                         max_position_embeddings=L, hidden_act="quick_gelu", pad_token_id=1, bos_token_id=0, eos_token_id=eos)
    torch.manual_seed(13)
    model = CLIPTextModel(cfg).eval().to(dtype).cuda()
    assert ClipTextEncoder.supports(model)
    gen = torch.Generator().manual_seed(4)
    n = 7
    lens = torch.randint(3, L + 1, (n,), generator=gen)
    lens[0] = L
    ids = torch.randint(3, vocab - 1, (n, L), generator=gen)
    am = (torch.arange(L)[None] < lens[:, None]).long()
    ids = torch.where(am.bool(), ids, torch.full_like(ids, 1))
    ids[:, 0] = 0
    ids[torch.arange(n), lens - 1] = vocab - 1 if eos == 2 else eos      # the EOS token closes every text (largest id of the vocabulary)
    ids, am = ids.cuda(), am.cuda()
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=am).pooler_output
    got = ClipTextEncoder(model).pooled(ids, am)
    assert got.shape == ref.shape
    assert _rel(got, ref) < tol


def test_unsupported_encoder_architecture_raises_instead_of_running_hf():
    """An encoder none of the HIP forwards cover must not run its HuggingFace forward silently (it did: round 2)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import mpt_args, tiny_clip_vision_config, tiny_opt_config
    from transformers import RobertaConfig
    from mmgl_amd.model import CrossAttentionModel
    rel = RobertaConfig(vocab_size=128, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, max_position_embeddings=40,
                        pad_token_id=1, type_vocab_size=1, position_embedding_type="relative_key")
    with pytest.raises(ValueError, match="no HIP forward"):
        CrossAttentionModel(mpt_args(context="all"), tokenizer=None, lm_config=tiny_opt_config(), text_config=rel, visual_config=tiny_clip_vision_config())
    args = mpt_args(context="all")
    args.allow_hf_encoder_forward = True            # round 3's opt-in is gone: there is no HF forward in the product to opt into
    with pytest.raises(ValueError, match="no HIP forward"):
        CrossAttentionModel(args, tokenizer=None, lm_config=tiny_opt_config(), text_config=rel, visual_config=tiny_clip_vision_config())
