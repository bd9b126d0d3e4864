"""Oracle (TEST INFRASTRUCTURE): functional fp32 restatement of the decoder-only LM
with interleaved tanh-gated cross-attention layers.

Reference: /root/reference/model/modelling_cross_attention.py
  masks                 :51-79, 455-476
  learned positions     :124-145
  MPTAttention.forward  :179-275
  MPTDecoderLayer       :304-375
  MPTDecoder.forward    :478-653
  MPTForCausalLM.forward:774-848

Parameters are passed as a flat dict ``p`` keyed exactly like the reference's
``MPTForCausalLM.state_dict()`` (``model.decoder.layers.0.self_attn.q_proj.weight``
...), so a golden fixture's state dict plugs in unchanged.  Everything here is
eval-mode (dropout = identity, LayerDrop off): the reference's RNG streams
cannot be reproduced, so parity is defined without them.
"""
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class LMConfig:
    vocab_size: int
    hidden_size: int
    num_attention_heads: int
    ffn_dim: int
    num_hidden_layers: int
    word_embed_proj_dim: int
    max_position_embeddings: int = 2048
    do_layer_norm_before: bool = True
    remove_final_layer_norm: bool = False
    activation_function: str = "relu"
    neighbor_layer_wise: int = 0          # 0 => no cross-attention layers
    flamingo: bool = True                  # peft_type == "flamingo"
    pad_token_id: int = 1


# ----------------------------------------------------------------------------- masks
def expand_mask(mask: torch.Tensor, dtype: torch.dtype, tgt_len: Optional[int] = None) -> torch.Tensor:
    """[B,S] {0,1}/bool -> additive [B,1,T,S]: 0 where valid, finfo(dtype).min where masked
    (reference :68-79)."""
    B, S = mask.shape
    T = S if tgt_len is None else tgt_len
    neg = torch.finfo(dtype).min
    add = torch.zeros(B, S, dtype=dtype)
    add[~mask.to(torch.bool)] = neg
    return add[:, None, None, :].expand(B, 1, T, S)


def causal_mask(B: int, T: int, dtype: torch.dtype) -> torch.Tensor:
    """additive lower-triangular mask [B,1,T,T] (reference :51-65)."""
    neg = torch.finfo(dtype).min
    m = torch.full((T, T), neg, dtype=dtype).triu(1)
    return m[None, None].expand(B, 1, T, T)


def decoder_self_mask(attention_mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """causal + key-padding, summed exactly as the reference does (:455-476).  NB the sum of two
    finfo.min overflows to -inf in fp32; the attention's max(., finfo.min) clamp repairs it."""
    B, T = attention_mask.shape
    return expand_mask(attention_mask, dtype, T) + causal_mask(B, T, dtype)


def learned_position_ids(attention_mask: torch.Tensor) -> torch.Tensor:
    """cumsum(mask)*mask - 1 + 2  (reference :135-145)."""
    m = attention_mask.long()
    return torch.cumsum(m, dim=1) * m - 1 + 2


# ----------------------------------------------------------------------------- attention
def attention_core(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, add_mask: Optional[torch.Tensor],
                   num_heads: int, head_mask: Optional[torch.Tensor] = None, keep: Optional[torch.Tensor] = None,
                   p_drop: float = 0.0, return_probs: bool = False):
    """softmax(max(QK^T + M, finfo.min)) V on [B,T,d] / [B,S,d] tensors; q is already scaled
    (reference :206-271).  Returns [B,T,d] with heads merged.
    Options of :237-256 (pinned by tests/golden/g10_attention_options_*.npz): `head_mask` [H] scales each head's probabilities
    (:237-244); `return_probs` also returns them as [B,H,T,S], head-masked and BEFORE dropout (:246-254); `keep` [B,H,T,S] (bool)
    with `p_drop` is the attention-probability dropout of :256 with an explicit mask, probs * keep / (1 - p_drop)."""
    B, T, d = q.shape
    S = k.shape[1]
    D = d // num_heads
    qh = q.reshape(B, T, num_heads, D).permute(0, 2, 1, 3)
    kh = k.reshape(B, S, num_heads, D).permute(0, 2, 1, 3)
    vh = v.reshape(B, S, num_heads, D).permute(0, 2, 1, 3)
    scores = qh @ kh.transpose(-1, -2)                      # [B,H,T,S]
    if add_mask is not None:
        scores = scores + add_mask
        # torch.maximum, not clamp: at an exact tie (every masked entry IS finfo.min after the add)
        # autograd splits the gradient 50/50 between the two operands, so the reference's dScores is
        # HALVED on masked entries.  Invisible on partially-masked rows (P = 0 there), but a
        # fully-masked sample gets 0.5x dQ / dK.  The HIP backward reproduces this.
        scores = torch.maximum(scores, torch.tensor(torch.finfo(scores.dtype).min, dtype=scores.dtype))
    probs = torch.softmax(scores, dim=-1)
    if head_mask is not None:
        probs = head_mask.view(1, -1, 1, 1) * probs
    weights = probs
    if keep is not None and p_drop > 0.0:
        probs = probs * keep.to(probs.dtype) / (1.0 - p_drop)
    out = probs @ vh                                         # [B,H,T,D]
    out = out.permute(0, 2, 1, 3).reshape(B, T, d)
    return (out, weights) if return_probs else out


def attention(p: Dict[str, torch.Tensor], pre: str, hidden: torch.Tensor, add_mask: Optional[torch.Tensor],
              num_heads: int, kv_source: Optional[torch.Tensor] = None, head_mask: Optional[torch.Tensor] = None,
              keep: Optional[torch.Tensor] = None, p_drop: float = 0.0, return_probs: bool = False):
    """MPTAttention.forward (:179-275).  kv_source=None => self-attention.  head_mask / keep / p_drop / return_probs: attention_core."""
    d = hidden.shape[-1]
    D = d // num_heads
    src = hidden if kv_source is None else kv_source
    q = F.linear(hidden, p[pre + "q_proj.weight"], p.get(pre + "q_proj.bias")) * (D ** -0.5)
    k = F.linear(src, p[pre + "k_proj.weight"], p.get(pre + "k_proj.bias"))
    v = F.linear(src, p[pre + "v_proj.weight"], p.get(pre + "v_proj.bias"))
    o = attention_core(q, k, v, add_mask, num_heads, head_mask, keep, p_drop, return_probs)
    if return_probs:
        return F.linear(o[0], p[pre + "out_proj.weight"], p.get(pre + "out_proj.bias")), o[1]
    return F.linear(o, p[pre + "out_proj.weight"], p.get(pre + "out_proj.bias"))


def _act(name: str):
    return {"relu": F.relu, "gelu": F.gelu}[name]


def _ln(p, pre, x):
    return F.layer_norm(x, (x.shape[-1],), p.get(pre + "weight"), p.get(pre + "bias"), 1e-5)


def decoder_layer(p: Dict[str, torch.Tensor], pre: str, hidden: torch.Tensor, self_mask: Optional[torch.Tensor],
                  cfg: LMConfig, neighbor_embeds: Optional[torch.Tensor] = None,
                  neighbor_mask: Optional[torch.Tensor] = None, cross: bool = False) -> torch.Tensor:
    """MPTDecoderLayer.forward (:304-375), eval mode."""
    gated = cross and cfg.flamingo
    g1 = torch.tanh(p[pre + "gating1"]) if gated else 1.0
    g2 = torch.tanh(p[pre + "gating2"]) if gated else 1.0

    res = hidden
    x = _ln(p, pre + "self_attn_layer_norm.", hidden) if cfg.do_layer_norm_before else hidden
    if cross:
        a = attention(p, pre + "self_attn.", x, neighbor_mask, cfg.num_attention_heads, kv_source=neighbor_embeds)
    else:
        a = attention(p, pre + "self_attn.", x, self_mask, cfg.num_attention_heads)
    hidden = res + g1 * a
    if not cfg.do_layer_norm_before:
        hidden = _ln(p, pre + "self_attn_layer_norm.", hidden)

    res = hidden
    x = _ln(p, pre + "final_layer_norm.", hidden) if cfg.do_layer_norm_before else hidden
    x = F.linear(x, p[pre + "fc1.weight"], p.get(pre + "fc1.bias"))
    x = _act(cfg.activation_function)(x)
    x = F.linear(x, p[pre + "fc2.weight"], p.get(pre + "fc2.bias"))
    hidden = res + g2 * x
    if not cfg.do_layer_norm_before:
        hidden = _ln(p, pre + "final_layer_norm.", hidden)
    return hidden


def decoder_forward(p: Dict[str, torch.Tensor], cfg: LMConfig, input_ids: torch.Tensor,
                    attention_mask: torch.Tensor, neighbor_embeds: Optional[torch.Tensor] = None,
                    neighbor_attention_mask: Optional[torch.Tensor] = None,
                    inputs_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MPTDecoder.forward (:478-653): embeddings, frozen layers with a gated cross-attention layer after
    every ``neighbor_layer_wise``-th one, final LN, project_out."""
    dec = "model.decoder."
    if inputs_embeds is None:
        inputs_embeds = F.embedding(input_ids, p[dec + "embed_tokens.weight"])
    B, T = inputs_embeds.shape[:2]
    dtype = inputs_embeds.dtype
    self_mask = decoder_self_mask(attention_mask, dtype)
    nmask = None
    if neighbor_attention_mask is not None:
        nmask = expand_mask(neighbor_attention_mask, dtype, T)
    pos = F.embedding(learned_position_ids(attention_mask), p[dec + "embed_positions.weight"])
    x = inputs_embeds
    if dec + "project_in.weight" in p:
        x = F.linear(x, p[dec + "project_in.weight"])
    h = x + pos
    wise = cfg.neighbor_layer_wise
    for i in range(cfg.num_hidden_layers):
        h = decoder_layer(p, f"{dec}layers.{i}.", h, self_mask, cfg)
        if wise and neighbor_embeds is not None and (i + 1) % wise == 0:
            k = (i + 1) // wise - 1
            h = decoder_layer(p, f"{dec}neighbor_layers.{k}.", h, self_mask, cfg,
                              neighbor_embeds=neighbor_embeds, neighbor_mask=nmask, cross=True)
    if cfg.do_layer_norm_before and not cfg.remove_final_layer_norm:
        h = _ln(p, dec + "final_layer_norm.", h)
    if dec + "project_out.weight" in p:
        h = F.linear(h, p[dec + "project_out.weight"])
    return h


def causal_lm_forward(p, cfg: LMConfig, input_ids, attention_mask, labels=None, neighbor_embeds=None,
                      neighbor_attention_mask=None, inputs_embeds=None):
    """MPTForCausalLM.forward (:774-848).  Loss = mean CE over ALL shifted positions (pads included,
    labels are real ids; -100 is honoured because F.cross_entropy's default ignore_index is -100)."""
    h = decoder_forward(p, cfg, input_ids, attention_mask, neighbor_embeds, neighbor_attention_mask,
                        inputs_embeds=inputs_embeds)
    logits = F.linear(h, p["lm_head.weight"])
    loss = None
    if labels is not None:
        V = logits.shape[-1]
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1))
    return logits, loss
