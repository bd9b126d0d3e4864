"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import torch

from helpers import Fixture, assert_close
from oracle import lm_ref, wrapper_ref

TOL = 2e-5   # fp32 restatement vs fp32 reference: only summation-order noise is allowed


def test_g4_attention_fwd_bwd():
    fx = Fixture("g4_attention.npz")
    H = fx.meta["H"]
    hidden = fx.inp["hidden"].clone().requires_grad_()
    ne = fx.inp["neighbor_embeds"].clone().requires_grad_()
    p = {k: v.clone().requires_grad_() for k, v in fx.p.items()}
    m4 = lm_ref.expand_mask(fx.inp["valid"], hidden.dtype, hidden.shape[1])
    out = lm_ref.attention(p, "", hidden, m4, H, kv_source=ne)
    assert_close(out, fx.out["out"], TOL, "attention out")
    assert torch.isfinite(out).all()          # fully-masked sample: uniform softmax, not NaN
    (out * fx.inp["w"]).sum().backward()
    assert_close(hidden.grad, fx.grad["hidden"], TOL, "d hidden")
    assert_close(ne.grad, fx.grad["neighbor_embeds"], TOL, "d neighbor_embeds")
    for k in p:
        assert_close(p[k].grad, fx.grad[k], TOL, f"d {k}")


def test_g4_fully_masked_row_is_uniform():
    fx = Fixture("g4_attention.npz")
    H, S = fx.meta["H"], fx.meta["S"]
    q = torch.randn(1, 5, 64)
    k = torch.randn(1, S, 64)
    v = torch.randn(1, S, 64)
    m4 = lm_ref.expand_mask(torch.zeros(1, S, dtype=torch.bool), q.dtype, 5)
    out = lm_ref.attention_core(q, k, v, m4, H)
    assert_close(out, v.mean(dim=1, keepdim=True).expand(1, 5, 64), 1e-6, "uniform")


def test_g10_attention_options_fwd_bwd():
    """The options of MPTAttention.forward :237-256 -- layer_head_mask, output_attentions, attention-probability dropout (with the
    fixture's explicit mask) -- on both call sites, against outputs of the reference itself."""
    for name in ("cross", "self"):
        fx = Fixture(f"g10_attention_options_{name}.npz")
        H, pd = fx.meta["H"], fx.meta["p_drop"]
        hidden = fx.inp["hidden"].clone().requires_grad_()
        p = {k: v.clone().requires_grad_() for k, v in fx.p.items()}
        T = hidden.shape[1]
        if name == "cross":
            ne = fx.inp["neighbor_embeds"].clone().requires_grad_()
            m4 = lm_ref.expand_mask(fx.inp["valid"], hidden.dtype, T)
        else:
            ne = None
            m4 = lm_ref.decoder_self_mask(fx.inp["valid"], hidden.dtype)
        out, w = lm_ref.attention(p, "", hidden, m4, H, kv_source=ne, head_mask=fx.inp["head_mask"], keep=fx.inp["keep"], p_drop=pd,
                                  return_probs=True)
        assert_close(out, fx.out["out"], TOL, f"{name} out")
        assert_close(w, fx.out["attn_weights"], TOL, f"{name} attn_weights")
        (out * fx.inp["w"]).sum().backward()
        assert_close(hidden.grad, fx.grad["hidden"], TOL, f"{name} d hidden")
        if ne is not None:
            assert_close(ne.grad, fx.grad["neighbor_embeds"], TOL, f"{name} d neighbor_embeds")
        for k in p:
            assert_close(p[k].grad, fx.grad[k], TOL, f"{name} d {k}")


def _layer_cfg(meta):
    return lm_ref.LMConfig(vocab_size=128, hidden_size=meta["d"], num_attention_heads=meta["H"], ffn_dim=meta["ffn"],
                           num_hidden_layers=1, word_embed_proj_dim=meta["d"], do_layer_norm_before=meta["pre_ln"])


def test_g5_layer_fwd_bwd():
    for name in ("g5_layer_preln.npz", "g5_layer_postln.npz"):
        fx = Fixture(name)
        cfg = _layer_cfg(fx.meta)
        hidden = fx.inp["hidden"].clone().requires_grad_()
        ne = fx.inp["neighbor_embeds"].clone().requires_grad_()
        p = {k: v.clone().requires_grad_() for k, v in fx.p.items()}
        m4 = lm_ref.expand_mask(fx.inp["valid"], hidden.dtype, hidden.shape[1])
        out = lm_ref.decoder_layer(p, "", hidden, None, cfg, neighbor_embeds=ne, neighbor_mask=m4, cross=True)
        assert_close(out, fx.out["out"], TOL, f"{name} out")
        (out * fx.inp["w"]).sum().backward()
        assert_close(hidden.grad, fx.grad["hidden"], TOL, "d hidden")
        assert_close(ne.grad, fx.grad["neighbor_embeds"], TOL, "d ne")
        for k in p:
            assert_close(p[k].grad, fx.grad[k], TOL, f"{name} d {k}")


def _lm_cfg(p, pre_ln, wise, heads=4):
    dec = "model.decoder."
    return lm_ref.LMConfig(
        vocab_size=p["lm_head.weight"].shape[0], hidden_size=p[dec + "layers.0.fc1.weight"].shape[1],
        num_attention_heads=heads, ffn_dim=p[dec + "layers.0.fc1.weight"].shape[0],
        num_hidden_layers=len({k.split(".")[3] for k in p if k.startswith(dec + "layers.")}),
        word_embed_proj_dim=p["lm_head.weight"].shape[1], do_layer_norm_before=pre_ln, neighbor_layer_wise=wise)


def test_g3_lm_without_neighbors_matches_reference_and_hf():
    for name in ("g3_lm_raw_preln.npz", "g3_lm_raw_postln_proj.npz"):
        fx = Fixture(name)
        cfg = _lm_cfg(fx.p, fx.meta["pre_ln"], 0)
        logits, loss = lm_ref.causal_lm_forward(fx.p, cfg, fx.inp["input_ids"], fx.inp["attention_mask"],
                                                fx.inp["labels"])
        assert_close(logits, fx.out["logits"], TOL, f"{name} logits")
        assert_close(logits, fx.out["hf_logits"], TOL, f"{name} vs HF OPT")
        assert_close(loss, fx.out["loss"], TOL, f"{name} loss")


def _run_wrapper(fx, p):
    lm = {k[3:]: v for k, v in p.items() if k.startswith("lm.")}
    cfg = _lm_cfg(lm, True, fx.meta["wise"])
    return wrapper_ref.cross_attention_model_forward(
        p, cfg, fx.inp, fx.out["text_last_hidden"], fx.out.get("visual_pooled"), fx.meta["context"],
        fx.meta["n_tokens"])


def test_g1_wrapper_logits_loss_grads():
    for name in ("g1_wrapper_all.npz", "g1_wrapper_text_only.npz"):
        fx = Fixture(name)
        p = {k: v.clone() for k, v in fx.p.items()}
        for k in fx.grad:
            p[k].requires_grad_()
        logits, loss = _run_wrapper(fx, p)
        assert_close(logits, fx.out["logits"], TOL, f"{name} logits")
        assert_close(loss, fx.out["loss"], TOL, f"{name} loss")
        loss.backward()
        for k in fx.grad:
            assert_close(p[k].grad, fx.grad[k], 1e-4, f"{name} d {k}")


def test_gates_zero_is_identity():
    """with tanh(0) gates the cross-attention layers are the identity."""
    fx = Fixture("g1_wrapper_all.npz")
    p = {k: v.clone() for k, v in fx.p.items()}
    for k in p:
        if k.endswith("gating1") or k.endswith("gating2"):
            p[k].zero_()
    logits_a, _ = _run_wrapper(fx, p)
    lm = {k[3:]: v for k, v in p.items() if k.startswith("lm.")}
    cfg = _lm_cfg(lm, True, 0)
    logits_b, _ = lm_ref.causal_lm_forward(lm, cfg, fx.inp["input_ids"], fx.inp["attention_mask"], fx.inp["labels"])
    assert_close(logits_a, logits_b, 1e-6, "gates=0")


def test_g8_gcn():
    fx = Fixture("g8_gcn.npz")
    out = wrapper_ref.gcn_forward(fx.p["w1.weight"], fx.p["w2.weight"], fx.inp["X"], fx.inp["adj"])
    assert_close(out, fx.out["out"], TOL, "gcn")
