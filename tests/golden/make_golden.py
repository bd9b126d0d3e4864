#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (build container only).

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

The reference's Python never travels to the GPU box: only the numeric fixtures written here do
(inputs, explicit state dicts, expected outputs / gradients).  No reference source is copied.
Fixture key layout inside each .npz:
    meta            JSON string (config / scalars)
    p/<name>        parameter tensors (reference state_dict names)
    in/<name>       inputs
    out/<name>      expected outputs
    grad/<name>     expected gradients of out/loss (or of the stated scalar) w.r.t. p/<name> or in/<name>
"""
import importlib.util
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def save(fname, meta, **groups):
    flat = {"meta": np.array(json.dumps(meta))}
    for g, d in groups.items():
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            flat[f"{g}/{k}"] = np.asarray(v)
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **flat)
    print(f"wrote {fname}: {os.path.getsize(path)/1024:.1f} KiB, {len(flat)} arrays")


# --------------------------------------------------------------------------- tiny HF configs
def tiny_opt_config(pre_ln=True, proj=None):
    from transformers import OPTConfig
    return OPTConfig(vocab_size=128, hidden_size=64, num_attention_heads=4, ffn_dim=128, num_hidden_layers=4,
                     max_position_embeddings=64, word_embed_proj_dim=proj or 64, do_layer_norm_before=pre_ln,
                     dropout=0.1, attention_dropout=0.0, pad_token_id=1, bos_token_id=2, eos_token_id=2,
                     init_std=0.08)


def tiny_roberta_config():
    from transformers import RobertaConfig
    return RobertaConfig(vocab_size=128, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                         intermediate_size=64, max_position_embeddings=40, pad_token_id=1, type_vocab_size=1,
                         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


def tiny_clip_vision_config():
    from transformers import CLIPVisionConfig
    return CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2,
                            image_size=32, patch_size=16)


def mpt_args(**kw):
    a = dict(context="all", neighbor_mode="cross_attention", n_text_tokens=2, n_visual_tokens=2,
             text_model="roberta-tiny", visual_model="clip-vit-tiny", max_output_length=8, freeze_lm=False,
             model_name_or_path="opt-tiny", peft_type="flamingo", lora_r=4, lora_alpha=1.0, lora_dropout=0.0,
             neighbor_layer_wise=2)
    a.update(kw)
    return SimpleNamespace(**a)


def set_gates(lm, base=0.5):
    with torch.no_grad():
        for i, layer in enumerate(lm.model.decoder.neighbor_layers):
            layer.gating1.fill_(base + 0.1 * i)
            layer.gating2.fill_(-(base - 0.2) - 0.1 * i)


def make_batch(g, B=2, Lin=16, Lout=8, Nt=3, Ni=2, Ln=12, V=128, img=32, ragged=True):
    T = Lin + Lout
    ids = torch.randint(3, V, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.long)
    for b in range(B):
        lp = int(torch.randint(4, Lin + 1, (1,), generator=g)) if ragged else Lin
        ls = int(torch.randint(2, Lout + 1, (1,), generator=g)) if ragged else Lout
        ids[b, lp:Lin] = 1
        am[b, lp:Lin] = 0
        ids[b, Lin + ls - 1] = 2
        ids[b, Lin + ls:] = 1
        am[b, Lin + ls:] = 0
    nids = torch.randint(3, V, (B, Nt, Ln), generator=g)
    nam = torch.ones(B, Nt, Ln, dtype=torch.long)
    npos = torch.zeros(B, Nt, dtype=torch.long)
    ipos = torch.zeros(B, Ni, dtype=torch.long)
    tloc = torch.zeros(B, Nt, dtype=torch.long)
    iloc = torch.zeros(B, Ni, dtype=torch.long)
    imgs = torch.zeros(B, Ni, 3, img, img)
    for b in range(B):
        nt = int(torch.randint(1, Nt + 1, (1,), generator=g)) if b else Nt - 1
        ni = int(torch.randint(0, Ni + 1, (1,), generator=g)) if b else 1
        for j in range(Nt):
            if j < nt:
                ln = int(torch.randint(3, Ln + 1, (1,), generator=g))
                nids[b, j, 0] = 0
                nids[b, j, ln:] = 1
                nam[b, j, ln:] = 0
                npos[b, j] = j + 1
            else:
                nids[b, j] = 1
                nids[b, j, 0] = 0
                nids[b, j, 1] = 2
                nam[b, j, 2:] = 0
        for j in range(ni):
            imgs[b, j] = torch.randn(3, img, img, generator=g)
            ipos[b, j] = j + 1
        # slot order: slot 0 = first text neighbor, then a random interleave of the remaining valid ones,
        # then padding text slots, then padding image slots (data.py:349-359, 444-454)
        kinds = ["t"] * (nt - 1) + ["i"] * ni
        perm = torch.randperm(len(kinds), generator=g).tolist()
        order = ["t"] + [kinds[q] for q in perm]
        ti = ii = 0
        for loc, kd in enumerate(order):
            if kd == "t":
                tloc[b, ti] = loc
                ti += 1
            else:
                iloc[b, ii] = loc
                ii += 1
        loc = len(order)
        for j in range(nt, Nt):
            tloc[b, j] = loc
            loc += 1
        for j in range(ni, Ni):
            iloc[b, j] = loc
            loc += 1
    return dict(input_ids=ids, attention_mask=am, labels=ids.clone(), neighbor_input_ids=nids,
                neighbor_attention_mask=nam, neighbor_pos_ids=npos, text_locations=tloc, neighbor_images=imgs,
                neighbor_images_pos_ids=ipos, image_locations=iloc)


# --------------------------------------------------------------------------- fixtures
def golden_attention(xa):
    """G4: MPTAttention cross branch alone, ragged + one fully-masked sample, fwd + dq-side grads."""
    torch.manual_seed(4)
    cfg = xa.MPTConfig(mpt_args(), tiny_opt_config())
    B, H, T, S, D = 3, 4, 16, 12, 16
    attn = xa.MPTAttention(cfg, cross_attention=True).eval()
    for prm in attn.parameters():
        torch.nn.init.normal_(prm, std=0.15)
    hidden = torch.randn(B, T, H * D, requires_grad=True)
    ne = torch.randn(B, S, H * D, requires_grad=True)
    valid = torch.ones(B, S, dtype=torch.bool)
    valid[0, 7:] = False
    valid[1, :] = False          # fully masked sample -> uniform softmax, finite
    valid[2, ::3] = False
    m4 = xa._expand_mask(valid, hidden.dtype, tgt_len=T)
    out, _, _ = attn(hidden, neighbor_embeds=ne, neighbor_attention_mask=m4)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    save("g4_attention.npz", dict(B=B, H=H, T=T, S=S, D=D, scalar="sum(out*w)"),
         p={k: v for k, v in attn.state_dict().items()},
         **{"in": dict(hidden=hidden, neighbor_embeds=ne, valid=valid, w=w)},
         out=dict(out=out),
         grad=dict(hidden=hidden.grad, neighbor_embeds=ne.grad,
                   **{k: v.grad for k, v in attn.named_parameters()}))


def golden_attention_options(xa):
    """G10: MPTAttention with the options of :237-256 switched on -- layer_head_mask, output_attentions, attention-probability dropout in
    training mode -- on both call sites (cross: key mask incl. a fully-masked sample; self: causal + right padding).  The dropout mask
    is fixed: nn.functional.dropout is replaced, for the duration of the reference's forward, by `x * keep / (1 - p)` with a saved
    Bernoulli `keep` (the reference draws its mask from torch's RNG stream, which no other implementation can replay); everything else
    -- where the mask is applied, in which layout, what is returned -- is the reference's own code."""
    import torch.nn.functional as F
    torch.manual_seed(10)
    cfg = xa.MPTConfig(mpt_args(), tiny_opt_config())
    H, D, p = 4, 16, 0.25
    real_dropout = F.dropout
    for name, cross, B, T, S in (("cross", True, 3, 16, 12), ("self", False, 2, 16, 16)):
        attn = xa.MPTAttention(cfg, cross_attention=cross)
        for prm in attn.parameters():
            torch.nn.init.normal_(prm, std=0.15)
        attn.dropout = p
        attn.train()
        hidden = torch.randn(B, T, H * D, requires_grad=True)
        head_mask = torch.tensor([1.0, 0.0, 0.5, 2.0])
        keep = (torch.rand(B * H, T, S) >= p)

        def fixed_dropout(x, p=0.5, training=True, inplace=False, _keep=keep, _p=p):
            assert training and tuple(x.shape) == tuple(_keep.shape) and abs(p - _p) < 1e-12
            return x * _keep.to(x.dtype) / (1.0 - _p)

        if cross:
            ne = torch.randn(B, S, H * D, requires_grad=True)
            valid = torch.ones(B, S, dtype=torch.bool)
            valid[0, 7:] = False
            valid[1, :] = False
            valid[2, ::3] = False
            m4 = xa._expand_mask(valid, hidden.dtype, tgt_len=T)
            kw = dict(neighbor_embeds=ne, neighbor_attention_mask=m4)
        else:
            ne = None
            valid = torch.ones(B, T, dtype=torch.bool)
            valid[1, 11:] = False
            m4 = xa._expand_mask(valid, hidden.dtype, tgt_len=T) + xa._make_causal_mask((B, T), hidden.dtype, device=hidden.device)
            kw = dict(attention_mask=m4)
        F.dropout = fixed_dropout
        torch.nn.functional.dropout = fixed_dropout
        try:
            out, attn_w, _ = attn(hidden, layer_head_mask=head_mask, output_attentions=True, **kw)
        finally:
            F.dropout = real_dropout
            torch.nn.functional.dropout = real_dropout
        w = torch.randn_like(out)
        (out * w).sum().backward()
        ins = dict(hidden=hidden, valid=valid, w=w, head_mask=head_mask, keep=keep.reshape(B, H, T, S))
        grads = dict(hidden=hidden.grad, **{k: v.grad for k, v in attn.named_parameters()})
        if cross:
            ins["neighbor_embeds"] = ne
            grads["neighbor_embeds"] = ne.grad
        save(f"g10_attention_options_{name}.npz", dict(B=B, H=H, T=T, S=S, D=D, p_drop=p, scalar="sum(out*w)"),
             p={k: v for k, v in attn.state_dict().items()}, **{"in": ins}, out=dict(out=out, attn_weights=attn_w), grad=grads)


def golden_layer(xa):
    """G5: MPTDecoderLayer(cross, flamingo) fwd+bwd, pre-LN and post-LN."""
    for tag, pre_ln in (("preln", True), ("postln", False)):
        torch.manual_seed(5)
        cfg = xa.MPTConfig(mpt_args(), tiny_opt_config(pre_ln=pre_ln))
        B, T, S, d = 2, 16, 10, 64
        layer = xa.MPTDecoderLayer(cfg, cross_attention=True).eval()
        for n_, prm in layer.named_parameters():
            if prm.dim() > 0:
                torch.nn.init.normal_(prm, std=0.12)
        with torch.no_grad():
            layer.gating1.fill_(0.6)
            layer.gating2.fill_(-0.4)
            layer.self_attn_layer_norm.weight.add_(0.9)
            layer.final_layer_norm.weight.add_(1.1)
        hidden = torch.randn(B, T, d, requires_grad=True)
        ne = torch.randn(B, S, d, requires_grad=True)
        valid = torch.ones(B, S, dtype=torch.bool)
        valid[0, 6:] = False
        valid[1, 1::2] = False
        m4 = xa._expand_mask(valid, hidden.dtype, tgt_len=T)
        out = layer(hidden, attention_mask=None, neighbor_embeds=ne, neighbor_attention_mask=m4)[0]
        w = torch.randn_like(out)
        (out * w).sum().backward()
        save(f"g5_layer_{tag}.npz", dict(B=B, T=T, S=S, d=d, H=4, ffn=128, pre_ln=pre_ln),
             p=layer.state_dict(), **{"in": dict(hidden=hidden, neighbor_embeds=ne, valid=valid, w=w)},
             out=dict(out=out),
             grad=dict(hidden=hidden.grad, neighbor_embeds=ne.grad,
                       **{k: v.grad for k, v in layer.named_parameters()}))


def golden_lm_raw(xa):
    """G3: fork without neighbors (neighbor_mode='raw') == HF OPTForCausalLM; stores the fork's logits."""
    from transformers import OPTForCausalLM
    for tag, pre_ln, proj in (("preln", True, None), ("postln_proj", False, 32)):
        torch.manual_seed(3)
        oc = tiny_opt_config(pre_ln=pre_ln, proj=proj)
        hf = OPTForCausalLM(oc).eval()
        mc = xa.MPTConfig(mpt_args(neighbor_mode="raw", peft_type="none"), oc)
        lm = xa.MPTForCausalLM(mc).eval()
        missing = lm.load_state_dict(hf.state_dict(), strict=False)
        lm.lm_head.weight = lm.model.decoder.embed_tokens.weight
        g = torch.Generator().manual_seed(33)
        b = make_batch(g)
        with torch.no_grad():
            o = lm(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
            ohf = hf(input_ids=b["input_ids"], attention_mask=b["attention_mask"], labels=b["labels"])
        diff = (o.logits - ohf.logits).abs().max().item()
        print(f"  fork(raw) vs HF OPT [{tag}] max|dlogits| = {diff:.2e}; missing={list(missing.missing_keys)[:3]}")
        assert diff < 1e-4
        save(f"g3_lm_raw_{tag}.npz", dict(pre_ln=pre_ln, proj=proj or 64, hf_max_abs_diff=diff),
             p=lm.state_dict(), **{"in": {k: b[k] for k in ("input_ids", "attention_mask", "labels")}},
             out=dict(logits=o.logits, loss=o.loss, hf_logits=ohf.logits))


def build_wrapper(xa, context):
    """Full CrossAttentionModel on tiny random HF models (from_pretrained patched)."""
    from transformers import OPTForCausalLM, RobertaModel, CLIPVisionModel
    oc, rc, vc = tiny_opt_config(), tiny_roberta_config(), tiny_clip_vision_config()
    saved = (xa.AutoConfig.from_pretrained, xa.AutoModelForCausalLM.from_pretrained,
             xa.RobertaModel.from_pretrained, xa.CLIPVisionModel.from_pretrained)
    xa.AutoConfig.from_pretrained = staticmethod(lambda *a, **k: oc)
    xa.AutoModelForCausalLM.from_pretrained = staticmethod(lambda *a, **k: OPTForCausalLM(oc))
    xa.RobertaModel.from_pretrained = staticmethod(lambda *a, **k: RobertaModel(rc, add_pooling_layer=False))
    xa.CLIPVisionModel.from_pretrained = staticmethod(lambda *a, **k: CLIPVisionModel(vc))
    try:
        w = xa.CrossAttentionModel(mpt_args(context=context), tokenizer=None)
    finally:
        (xa.AutoConfig.from_pretrained, xa.AutoModelForCausalLM.from_pretrained,
         xa.RobertaModel.from_pretrained, xa.CLIPVisionModel.from_pretrained) = saved
    w.lm.lm_head.weight = w.lm.model.decoder.embed_tokens.weight     # tie
    set_gates(w.lm)
    w.eval()
    return w, oc, rc, vc


def golden_wrapper(xa):
    """G1/G2: whole CrossAttentionModel (context=all / text_only): logits, loss, grads of trainables."""
    for tag, context in (("all", "all"), ("text_only", "text_only")):
        torch.manual_seed(1)
        w, oc, rc, vc = build_wrapper(xa, context)
        g = torch.Generator().manual_seed(11)
        b = make_batch(g)
        kw = dict(b)
        if context == "text_only":
            for k in ("neighbor_images", "neighbor_images_pos_ids", "image_locations"):
                kw.pop(k)
        o = w(**kw)
        o.loss.backward()
        # encoder outputs, so the oracle can be checked without HF modules too
        with torch.no_grad():
            B, Nt, Ln = b["neighbor_input_ids"].shape
            tl = w.text_model(input_ids=b["neighbor_input_ids"].reshape(-1, Ln),
                              attention_mask=b["neighbor_attention_mask"].reshape(-1, Ln)).last_hidden_state
            extra = dict(text_last_hidden=tl)
            if context == "all":
                extra["visual_pooled"] = w.visual_model(b["neighbor_images"].reshape(-1, 3, 32, 32)).pooler_output
        grads = {k: v.grad for k, v in w.named_parameters() if v.requires_grad and v.grad is not None}
        nograd = [k for k, v in w.named_parameters() if v.requires_grad and v.grad is None]
        trainable = sorted(k for k, v in w.named_parameters() if v.requires_grad)
        meta = dict(context=context, n_tokens=2, wise=2, trainable=trainable, nograd=nograd,
                    opt=dict(vocab_size=128, hidden_size=64, heads=4, ffn=128, layers=4, max_pos=64),
                    roberta=dict(vocab_size=128, hidden_size=32, layers=2, heads=2, inter=64, max_pos=40),
                    clip=dict(hidden_size=32, inter=64, layers=2, heads=2, image=32, patch=16))
        save(f"g1_wrapper_{tag}.npz", meta, p=w.state_dict(), **{"in": b}, out=dict(logits=o.logits, loss=o.loss, **extra),
             grad=grads)


def golden_gcn():
    g = _load("ref_graph", f"{REF}/model/graph.py")
    torch.manual_seed(8)
    net = g.GCN(input_dim=12, output_dim=12, hidden_dim=7)
    X = torch.randn(2, 5, 12)
    adj = torch.rand(2, 6, 6)
    out = net(X, adj)
    save("g8_gcn.npz", {}, p=net.state_dict(), **{"in": dict(X=X, adj=adj)}, out=dict(out=out))


def golden_cider():
    pkg = types.ModuleType("ref_cider")
    pkg.__path__ = [f"{REF}/wikiweb2m/cider"]
    sys.modules["ref_cider"] = pkg
    _load("ref_cider.cider_scorer", f"{REF}/wikiweb2m/cider/cider_scorer.py")
    c = _load("ref_cider.cider", f"{REF}/wikiweb2m/cider/cider.py")
    cases = {
        "toy3": (["the cat sat on the mat", "a dog barks at the mailman", "birds fly south in winter"],
                 ["the cat sat on a mat", "a dog barks", "birds fly south in winter"]),
        "mixed": (["the tower was completed in 1889 and is 330 metres tall", "it is a species of frog",
                   "the river flows through three countries", "he won the election in 2004",
                   "the album was released by the band in june"],
                  ["the tower is 330 metres tall", "a species of frog", "completely unrelated words here",
                   "he won the election in 2004", ""]),
    }
    out = {}
    for name, (refs, hyps) in cases.items():
        gts = {i: [r] for i, r in enumerate(refs)}
        res = {i: [h] for i, h in enumerate(hyps)}
        score, scores = c.Cider().compute_score(gts, res)
        out[name] = dict(refs=refs, hyps=hyps, score=float(score), scores=[float(s) for s in scores])
        print(f"  cider[{name}] = {score!r}")
    with open(os.path.join(HERE, "g7_cider.json"), "w") as f:
        json.dump(out, f, indent=1)


def main():
    torch.set_num_threads(4)
    xa = _load("ref_xattn", f"{REF}/model/modelling_cross_attention.py")
    golden_attention(xa)
    golden_attention_options(xa)
    golden_layer(xa)
    golden_lm_raw(xa)
    golden_wrapper(xa)
    golden_gcn()
    golden_cider()
    import make_golden_host
    make_golden_host.main()


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
