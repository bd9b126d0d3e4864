"""Decoder-only LM with interleaved tanh-gated neighbor cross-attention layers, MI355X-native.

Mirrors the module API of the reference's model/modelling_cross_attention.py (class names, constructor
arguments, forward signatures, state_dict key names, ValueErrors) so `run_generation` and model checkpoints
interchange, and every op of the hot path runs through the hand-written HIP kernels of libmmgl_hip.so (mmgl_amd.ops):
  * gated cross-attention layers (trainable): fused-epilogue MFMA projections, the single-pass masked cross-attention
    core, LayerNorm, the gated residual(+dropout);
  * frozen OPT layers: one fused-QKV GEMM, causal flash attention reading Q/K/V in place, residual add folded into the
    following LayerNorm, FFN GEMMs with the ReLU in the epilogue and its backward in the epilogue of fc2's dgrad --
    all GEMMs on the persistent ping-pong MFMA kernel (csrc/gemm8p.hip), dgrads against cached W^T copies;
  * lm_head + shifted token cross-entropy, the interleave scatter, learned positions;
  * frozen RoBERTa / CLIP encoders: packed (padding-free) forward in encoders.py on the same kernels.

Deliberate deviations from the reference (all documented in DESIGN.md):
  * `args.neighbor_layer_wise` is optional: default num_hidden_layers // num_neighbor_layers (:92 reads an
    attribute `Arguments` never defines).
  * neighbor_mode "embedding" together with peft_type "flamingo" selects this cross-attention path, as the
    README pairs them; "cross_attention" is accepted too (:433, :1072, :1080 vs data.py:167).
  * the interleave buffer is allocated in the compute dtype on the device (:1095 allocates fp32 on the host).
  * `train()` returns self (:1029-1036 returns None).
  * masks are never materialised: cross-attention takes the [B,S] key mask (:545-546 builds [B,1,T,S]), self-attention
    the [B,T] key mask with causality implied (:455-476 builds [B,1,T,T]).  What the additive masks can express and the
    kernels cannot (per-head masks, attention weights as an output, attention dropout, a sample whose first key is
    masked) raises ValueError instead of taking a slower path.
GPU only: there is no CPU fallback and no alternative backend (ops raise if tensors are not on the device).
"""
import math
import os
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import (AutoConfig, AutoModelForCausalLM, CLIPTextModel, CLIPVisionModel, PretrainedConfig,
                          RobertaModel)
from transformers.activations import ACT2FN
from transformers.modeling_outputs import BaseModelOutputWithPast, CausalLMOutputWithPast

from .. import ops
from .encoders import ClipTextEncoder, PackedTextEncoder, PackedVisionEncoder

CROSS_MODES = ("cross_attention", "embedding")


def uses_cross_attention(neighbor_mode: str, peft_type: str = "flamingo") -> bool:
    return neighbor_mode == "cross_attention" or (neighbor_mode == "embedding" and peft_type == "flamingo")


# ----------------------------------------------------------------------------------------------- masks
def _make_causal_mask(input_ids_shape, dtype, device, past_key_values_length: int = 0):
    """Additive causal mask [B,1,T,T+past] (reference :51-65)."""
    bsz, tgt_len = input_ids_shape
    neg = torch.finfo(dtype).min
    mask = torch.full((tgt_len, tgt_len), neg, dtype=dtype, device=device).triu_(1)
    if past_key_values_length > 0:
        mask = torch.cat([torch.zeros(tgt_len, past_key_values_length, dtype=dtype, device=device), mask], dim=-1)
    return mask[None, None].expand(bsz, 1, tgt_len, tgt_len + past_key_values_length)


def _expand_mask(mask, dtype, tgt_len: Optional[int] = None):
    """[B,S] -> additive [B,1,T,S] (reference :68-79).  Only the frozen self-attention layers use it."""
    bsz, src_len = mask.shape
    tgt_len = src_len if tgt_len is None else tgt_len
    add = torch.zeros(bsz, src_len, dtype=dtype, device=mask.device)
    add.masked_fill_(~mask.to(torch.bool), torch.finfo(dtype).min)
    return add[:, None, None, :].expand(bsz, 1, tgt_len, src_len)


def _key_valid_from(mask: torch.Tensor) -> torch.Tensor:
    """Accept the [B,S] key mask (preferred) or the reference's additive [B,1,T,S] mask."""
    if mask.dim() == 2:
        return mask
    if mask.dim() == 4 and mask.shape[1] == 1:
        return mask[:, 0, 0, :] == 0
    raise ValueError(f"neighbor_attention_mask should be [bsz, src_len] or [bsz, 1, tgt_len, src_len], but is {tuple(mask.shape)}")


class MPTConfig(PretrainedConfig):
    """OPT config + the MMGL knobs (reference :82-121)."""

    def __init__(self, args=None, opt_config=None, **kwargs):
        if opt_config is None:          # transformers re-instantiates configs with kwargs only
            super().__init__(**kwargs)
            return
        super().__init__(pad_token_id=opt_config.pad_token_id, bos_token_id=opt_config.bos_token_id,
                         eos_token_id=opt_config.eos_token_id, **kwargs)
        n_layers = opt_config.num_hidden_layers
        wise = getattr(args, "neighbor_layer_wise", None)
        if wise is None:
            n_nb = max(1, int(getattr(args, "num_neighbor_layers", 4)))
            wise = max(1, n_layers // n_nb)
        self.neighbor_layer_wise = int(wise)
        self.neighbor_mode = args.neighbor_mode
        self.peft_type = args.peft_type
        self.lora_r = getattr(args, "lora_r", 64)
        self.lora_alpha = getattr(args, "lora_alpha", 1)
        self.lora_dropout = getattr(args, "lora_dropout", 0.0)
        for name in ("vocab_size", "max_position_embeddings", "num_attention_heads", "word_embed_proj_dim", "ffn_dim",
                     "hidden_size", "num_hidden_layers", "dropout", "attention_dropout", "activation_function", "init_std",
                     "layerdrop", "use_cache", "do_layer_norm_before", "enable_bias", "layer_norm_elementwise_affine",
                     "_remove_final_layer_norm"):
            setattr(self, name, getattr(opt_config, name))


class MPTLearnedPositionalEmbedding(nn.Embedding):
    """Learned positions with OPT's +2 offset; ids come from the HIP scan kernel (reference :124-145)."""

    def __init__(self, num_embeddings: int, embedding_dim: int):
        self.offset = 2
        super().__init__(num_embeddings + self.offset, embedding_dim)

    def forward(self, attention_mask: torch.LongTensor, past_key_values_length: int = 0):
        positions = ops.position_ids(attention_mask)          # already includes the +2 offset
        positions = positions[:, past_key_values_length:]
        return F.embedding(positions, self.weight)


def _lin(module: nn.Module, x, **kw):
    """nn.Linear on the HIP GEMMs: frozen parameters -> ops.frozen_linear (dgrad only, cached W^T), else ops.linear.
    Anything that is not a plain nn.Linear (a LoRALinear wrapper, model/modelling_self_attention.py) runs its own forward."""
    if type(module) is not nn.Linear:
        return module(x)
    if module.weight.requires_grad or (module.bias is not None and module.bias.requires_grad):
        return ops.linear(x, module.weight, module.bias)
    return ops.frozen_linear(x, module.weight, module.bias, **kw)


class MPTAttention(nn.Module):
    """Multi-head attention; cross_attention=True attends over the neighbor tokens (reference :148-275)."""

    def __init__(self, config, cross_attention):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.dropout = config.attention_dropout
        self.head_dim = self.embed_dim // self.num_heads
        bias = config.enable_bias
        if self.head_dim * self.num_heads != self.embed_dim:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {self.embed_dim}"
                             f" and `num_heads`: {self.num_heads}).")
        self.scaling = self.head_dim ** -0.5
        self.is_decoder = False
        self.k_proj = nn.Linear(self.embed_dim, self.embed_dim, bias=bias)
        self.q_proj = nn.Linear(self.embed_dim, self.embed_dim, bias=bias)
        self.out_proj = nn.Linear(self.embed_dim, self.embed_dim, bias=bias)
        self.cross_attention = cross_attention
        self.peft_type = config.peft_type
        self.v_proj = nn.Linear(self.embed_dim, self.embed_dim, bias=bias)

    # -- fused HIP path: projections with bias/scale epilogue + single-pass masked core
    def _forward_cross(self, hidden_states, neighbor_embeds, neighbor_attention_mask, layer_head_mask, output_attentions):
        if neighbor_embeds is None:
            raise ValueError("cross-attention layer called without neighbor_embeds")
        # layer_head_mask / output_attentions / attention-probability dropout (reference :237-256): the general HIP core
        # (ops.attn_general, exact and unfused); everything else -- every BASELINE config -- the fused kernel
        general = layer_head_mask is not None or output_attentions or (self.training and self.dropout > 0)
        bsz, tgt_len, _ = hidden_states.shape
        src_len = neighbor_embeds.shape[1]
        if neighbor_attention_mask is None:
            key_valid = torch.ones(bsz, src_len, dtype=torch.uint8, device=hidden_states.device)
        else:
            key_valid = _key_valid_from(neighbor_attention_mask)
            if key_valid.shape != (bsz, src_len):
                raise ValueError(f"Attention mask should be of size {(bsz, 1, tgt_len, src_len)}, but is"
                                 f" {tuple(neighbor_attention_mask.shape)}")
        q = ops.linear(hidden_states, self.q_proj.weight, self.q_proj.bias, out_scale=self.scaling)
        k = ops.linear(neighbor_embeds, self.k_proj.weight, self.k_proj.bias)
        v = ops.linear(neighbor_embeds, self.v_proj.weight, self.v_proj.bias)
        attn_w = None
        if general:
            o, attn_w = ops.attn_general(q, k, v, key_valid, self.num_heads, causal=False, head_mask=layer_head_mask, p_drop=self.dropout,
                                         training=self.training, output_attentions=output_attentions)
        else:
            o = ops.xattn_core(q, k, v, key_valid, self.num_heads)
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias), attn_w, None

    def _frozen_qkv(self):
        """[3d, d] weight / [3d] bias = (q_proj * scaling | k_proj | v_proj) when the three projections are frozen (the
        reference freezes the whole LM outside the cross-attention layers, :731-737), else None.  A derived copy: the
        module's own parameters and state_dict stay as loaded; rebuilt when they change (load_state_dict, .bfloat16())."""
        if any(type(m) is not nn.Linear for m in (self.q_proj, self.k_proj, self.v_proj)):
            return None                      # adapted projections (LoRA): each one runs its own forward
        ps = (self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.q_proj.bias, self.k_proj.bias, self.v_proj.bias)
        if any(p is None or p.requires_grad for p in ps):
            return None
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in ps)
        cache = self.__dict__.get("_qkv_cache")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.cat([ps[0].float() * self.scaling, ps[1].float(), ps[2].float()], 0).to(ps[0].dtype).contiguous()
                b = torch.cat([ps[3].float() * self.scaling, ps[4].float(), ps[5].float()], 0).to(ps[0].dtype).contiguous()
            cache = (key, (w, b))
            self.__dict__["_qkv_cache"] = cache
        return cache[1]

    def _lora_qkv(self):
        """The fused frozen [3d, d] weight / [3d] bias under LoRA adapters on q_proj and v_proj (peft's OPT targets) with a plain
        frozen k_proj, else None.  Same derived-copy cache as _frozen_qkv."""
        q, k, v = self.q_proj, self.k_proj, self.v_proj
        if not (hasattr(q, "lora_A") and hasattr(v, "lora_A") and type(k) is nn.Linear) or q.r != v.r or q.scaling != v.scaling:
            return None
        if self.training and (getattr(q, "lora_dropout", 0.0) > 0.0 or getattr(v, "lora_dropout", 0.0) > 0.0):
            return None                      # lora_dropout: each adapted projection runs its own (unfused) forward
        ps = (q.base_layer.weight, k.weight, v.base_layer.weight, q.base_layer.bias, k.bias, v.base_layer.bias)
        if any(p is None or p.requires_grad for p in ps):
            return None
        key = tuple((p.data_ptr(), p._version, p.dtype) for p in ps)
        cache = self.__dict__.get("_lora_qkv_cache")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = torch.cat([ps[0].float() * self.scaling, ps[1].float(), ps[2].float()], 0).to(ps[0].dtype).contiguous()
                b = torch.cat([ps[3].float() * self.scaling, ps[4].float(), ps[5].float()], 0).to(ps[0].dtype).contiguous()
            cache = (key, (w, b))
            self.__dict__["_lora_qkv_cache"] = cache
        return cache[1]

    # -- causal self-attention of the (frozen) OPT layers: HIP flash kernels, no [B,1,T,T] mask, no [B,H,T,T] scores
    def _forward_self(self, hidden_states, attention_mask, layer_head_mask, output_attentions, past_key_value=None):
        H = self.num_heads
        if attention_mask is None or attention_mask.dim() != 2:
            raise ValueError("self-attention takes the [bsz, seq_len] key mask (causality is implied); additive 4-D masks are "
                             f"not materialised on this path (got {None if attention_mask is None else tuple(attention_mask.shape)})")
        general = layer_head_mask is not None or output_attentions or (self.training and self.dropout > 0)
        if general and past_key_value is not None:
            raise ValueError("layer_head_mask / output_attentions / attention dropout together with a key / value prefix (prefix tuning) "
                             "are not implemented")
        fused = self._frozen_qkv()
        if general:
            # reference :237-256 on the self-attention call site: separate q / k / v (the fused-QKV node has no general core), the
            # general HIP core with the causal AND key mask
            d = hidden_states.shape[-1]
            if fused is not None:
                qkv = ops.frozen_linear(hidden_states, *fused)
                q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            else:
                q = _lin(self.q_proj, hidden_states) * self.scaling
                k, v = _lin(self.k_proj, hidden_states), _lin(self.v_proj, hidden_states)
            o, attn_w = ops.attn_general(q, k, v, attention_mask, H, causal=True, head_mask=layer_head_mask, p_drop=self.dropout,
                                         training=self.training, output_attentions=output_attentions)
            return _lin(self.out_proj, o), attn_w, None
        if past_key_value is not None:
            # prefix tuning (peft hands the learned per-layer key/value prefix to HF as past_key_values): P extra keys / values in
            # front of the layer's own, visible to every query; attention_mask is the [bsz, P + seq_len] key mask
            kp, vp = past_key_value
            B, T, d = hidden_states.shape
            P = kp.shape[-2]
            if fused is not None:
                qkv = ops.frozen_linear(hidden_states, *fused)
                q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
            else:
                q = _lin(self.q_proj, hidden_states) * self.scaling
                k, v = _lin(self.k_proj, hidden_states), _lin(self.v_proj, hidden_states)
            k = torch.cat([kp.to(k.dtype).expand(B, P, d), k], dim=1)
            v = torch.cat([vp.to(v.dtype).expand(B, P, d), v], dim=1)
            o = ops.selfattn_core_prefix(q, k, v, attention_mask, H, P)
        elif fused is not None:              # frozen layer: one QKV GEMM forward, one dgrad GEMM backward, no gradient adds
            o = ops.selfattn_core_fused(ops.frozen_linear(hidden_states, *fused), attention_mask, H)
        elif (lq := self._lora_qkv()) is not None and ops.lora_qkv_supported(hidden_states, lq[0], self.q_proj.r):
            # LoRA on q_proj / v_proj: one node for the three projections (fused base GEMM, one dgrad GEMM, no gradient adds)
            qkv = ops.lora_qkv(hidden_states, lq[0], lq[1], self.q_proj.lora_A, self.q_proj.lora_B, self.v_proj.lora_A, self.v_proj.lora_B,
                               self.q_proj.scaling, self.scaling)
            o = ops.selfattn_core_fused(qkv, attention_mask, H)
        else:
            if type(self.q_proj) is nn.Linear and self.q_proj.weight.requires_grad:
                q = ops.linear(hidden_states, self.q_proj.weight, self.q_proj.bias, out_scale=self.scaling)
            elif type(self.q_proj) is not nn.Linear and hasattr(self.q_proj, "lora_A"):
                q = self.q_proj(hidden_states, out_scale=self.scaling)           # LoRA: the scaling rides in the GEMM epilogues
            else:
                q = _lin(self.q_proj, hidden_states) * self.scaling
            o = ops.selfattn_core(q, _lin(self.k_proj, hidden_states), _lin(self.v_proj, hidden_states), attention_mask, H)
        return _lin(self.out_proj, o), None, None

    def forward(self, hidden_states, attention_mask=None, neighbor_embeds=None, neighbor_attention_mask=None,
                past_key_value=None, layer_head_mask=None, output_attentions=False):
        """Input shape: Batch x Time x Channel.  Returns (attn_output, attn_weights-or-None, None)."""
        if self.cross_attention:
            return self._forward_cross(hidden_states, neighbor_embeds, neighbor_attention_mask, layer_head_mask, output_attentions)
        return self._forward_self(hidden_states, attention_mask, layer_head_mask, output_attentions, past_key_value)




class _Deferred:
    """A layer output whose last residual add (`hidden = residual + branch`, reference :358-359) has not been done yet:
    the next pre-LN layer (or the final LayerNorm) performs it inside its first LayerNorm kernel (ops.add_layer_norm_pair).
    Internal to MPTDecoder's layer loop; anything else sees tensors (`_materialize`)."""
    __slots__ = ("residual", "branch", "p_drop", "training")

    def __init__(self, residual, branch, p_drop=0.0, training=False):
        self.residual, self.branch, self.p_drop, self.training = residual, branch, p_drop, training


def _materialize(h, *_):
    return ops.gated_residual(h.residual, h.branch, None, h.p_drop, h.training) if isinstance(h, _Deferred) else h


class MPTDecoderLayer(nn.Module):
    """OPT decoder layer; with cross_attention=True the Flamingo-style gated block (reference :278-375)."""

    def __init__(self, config, cross_attention=False):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.self_attn = MPTAttention(config, cross_attention)
        self.do_layer_norm_before = config.do_layer_norm_before
        self.dropout = config.dropout
        self.activation_name = config.activation_function
        self.activation_fn = ACT2FN[config.activation_function]
        affine = config.layer_norm_elementwise_affine
        self.self_attn_layer_norm = nn.LayerNorm(self.embed_dim, elementwise_affine=affine)
        self.fc1 = nn.Linear(self.embed_dim, config.ffn_dim, bias=config.enable_bias)
        self.fc2 = nn.Linear(config.ffn_dim, self.embed_dim, bias=config.enable_bias)
        self.final_layer_norm = nn.LayerNorm(self.embed_dim, elementwise_affine=affine)
        self.cross_attention = cross_attention
        self.peft_type = config.peft_type
        if self.cross_attention and self.peft_type == "flamingo":
            self.tanh_layer1 = nn.Tanh()
            self.tanh_layer2 = nn.Tanh()
            self.gating1 = nn.Parameter(torch.tensor(0.0))
            self.gating2 = nn.Parameter(torch.tensor(0.0))

    def _ln(self, ln, x):
        return ops.layer_norm(x, ln.weight, ln.bias, ln.eps)

    def _ln_fanout(self, ln, x):
        """(residual, LayerNorm(x)): ops.layer_norm_fanout when gradients flow (its backward adds the residual stream's gradient
        in the LayerNorm kernel), the plain kernel otherwise."""
        if torch.is_grad_enabled() and x.requires_grad:
            return ops.layer_norm_fanout(x, ln.weight, ln.bias, ln.eps)
        return x, ops.layer_norm(x, ln.weight, ln.bias, ln.eps)

    def _forward_cross(self, h, neighbor_embeds, neighbor_attention_mask, layer_head_mask, output_attentions):
        gated = self.peft_type == "flamingo"
        # pre-LN: the block input feeds the LayerNorm AND the residual add -- both gradients meet inside the LayerNorm backward kernel
        residual, x = self._ln_fanout(self.self_attn_layer_norm, h) if self.do_layer_norm_before else (h, h)
        a, attn_w, _ = self.self_attn(x, neighbor_embeds=neighbor_embeds, neighbor_attention_mask=neighbor_attention_mask,
                                      layer_head_mask=layer_head_mask, output_attentions=output_attentions)
        h = ops.gated_residual(residual, a, self.gating1 if gated else None, self.dropout, self.training)
        if not self.do_layer_norm_before:
            h = self._ln(self.self_attn_layer_norm, h)
        residual, x = self._ln_fanout(self.final_layer_norm, h) if self.do_layer_norm_before else (h, h)
        if self.activation_name == "relu":
            # fc1's ReLU backward rides in the epilogue of fc2's dgrad GEMM (mask_dx) instead of a separate pass over [M, ffn]
            x = ops.linear(x, self.fc1.weight, self.fc1.bias, act="relu", bwd_premasked=True)
            x = ops.linear(x, self.fc2.weight, self.fc2.bias, mask_dx=True)
        else:
            x = self.activation_fn(ops.linear(x, self.fc1.weight, self.fc1.bias))
            x = ops.linear(x, self.fc2.weight, self.fc2.bias)
        h = ops.gated_residual(residual, x, self.gating2 if gated else None, self.dropout, self.training)
        if not self.do_layer_norm_before:
            h = self._ln(self.final_layer_norm, h)
        return h, attn_w

    def _forward_self(self, h, attention_mask, layer_head_mask, output_attentions, defer_residual=False, past_key_value=None):
        """OPT layer (frozen in every peft mode of the reference, :731-737): GEMMs, attention (ops.selfattn_core*), LayerNorm and
        dropout + residual all run on this repo's HIP kernels.  Every `residual + dropout(branch)` is folded into the LayerNorm that
        follows it (ops.add_layer_norm_pair: one forward and one backward kernel per pair); the layer's last add can be
        left to the next layer's first LayerNorm (`defer_residual`, see _Deferred)."""
        pre = self.do_layer_norm_before
        ln1, ln2 = self.self_attn_layer_norm, self.final_layer_norm
        pair = lambda x, r, ln: ops.add_layer_norm_pair(x, r, ln.weight, ln.bias, ln.eps, self.dropout, self.training)
        if isinstance(h, _Deferred):
            if pre:
                h, x = ops.add_layer_norm_pair(h.branch, h.residual, ln1.weight, ln1.bias, ln1.eps, h.p_drop, h.training)
            else:
                h = x = _materialize(h)
        elif pre:
            h, x = self._ln_fanout(ln1, h)
        else:
            x = h
        residual = h
        a, attn_w, _ = self.self_attn(x, attention_mask=attention_mask, layer_head_mask=layer_head_mask,
                                      output_attentions=output_attentions, past_key_value=past_key_value)
        if pre:
            h, x = pair(a, residual, ln2)
        else:
            _, h = pair(a, residual, ln1)                                                   # post-LN: h = LN(residual + a)
            x = h
        residual = h
        if self.activation_name == "relu":
            # fc1's ReLU backward rides in the epilogue of fc2's dgrad GEMM (mask_dx): no pass over [M, ffn], nothing kept twice
            frozen = not any(p.requires_grad for p in (*self.fc1.parameters(), *self.fc2.parameters()))
            lin = ops.frozen_linear if frozen else ops.linear
            kw = dict(relu=True) if frozen else dict(act="relu")
            x = lin(lin(x, self.fc1.weight, self.fc1.bias, bwd_premasked=True, **kw), self.fc2.weight, self.fc2.bias, mask_dx=True)
        else:
            x = _lin(self.fc2, self.activation_fn(_lin(self.fc1, x)))
        if pre and defer_residual:
            return _Deferred(residual, x, self.dropout, self.training), attn_w
        if not pre:
            _, h = pair(x, residual, ln2)
            return h, attn_w
        return ops.gated_residual(residual, x, None, self.dropout, self.training), attn_w

    def forward(self, hidden_states, attention_mask=None, neighbor_embeds=None, neighbor_attention_mask=None,
                layer_head_mask=None, past_key_value=None, output_attentions=False, use_cache=False, defer_residual=False):
        if self.cross_attention:
            h, attn_w = self._forward_cross(_materialize(hidden_states), neighbor_embeds, neighbor_attention_mask, layer_head_mask,
                                            output_attentions)
        else:
            h, attn_w = self._forward_self(hidden_states, attention_mask, layer_head_mask, output_attentions, defer_residual, past_key_value)
        outputs = (h,)
        if output_attentions:
            outputs += (attn_w,)
        if use_cache:
            outputs += (None,)
        return outputs


class MPTPreTrainedModel(nn.Module):
    """Plain nn.Module base (the reference derives from transformers.PreTrainedModel only for init/config)."""
    config_class = MPTConfig
    base_model_prefix = "model"

    def __init__(self, config):
        super().__init__()
        self.config = config

    def _init_weights(self, module):
        std = self.config.init_std
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()

    def post_init(self):
        self.apply(self._init_weights)


class MPTDecoder(MPTPreTrainedModel):
    """Frozen OPT stack with a gated cross-attention layer after every `neighbor_layer_wise`-th layer (reference :400-653)."""

    def __init__(self, config: MPTConfig):
        super().__init__(config)
        self.dropout = config.dropout
        self.layerdrop = config.layerdrop
        self.padding_idx = config.pad_token_id
        self.max_target_positions = config.max_position_embeddings
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.word_embed_proj_dim, self.padding_idx)
        self.embed_positions = MPTLearnedPositionalEmbedding(config.max_position_embeddings, config.hidden_size)
        proj = config.word_embed_proj_dim != config.hidden_size
        self.project_out = nn.Linear(config.hidden_size, config.word_embed_proj_dim, bias=False) if proj else None
        self.project_in = nn.Linear(config.word_embed_proj_dim, config.hidden_size, bias=False) if proj else None
        if config.do_layer_norm_before and not config._remove_final_layer_norm:
            self.final_layer_norm = nn.LayerNorm(config.hidden_size, elementwise_affine=config.layer_norm_elementwise_affine)
        else:
            self.final_layer_norm = None
        self.cross_attention = uses_cross_attention(config.neighbor_mode, config.peft_type)
        self.neighbor_layer_wise = config.neighbor_layer_wise
        self.peft_type = config.peft_type
        self.layers = nn.ModuleList()
        self.neighbor_layers = nn.ModuleList()
        for l in range(config.num_hidden_layers):
            self.layers.append(MPTDecoderLayer(config))
            if self.cross_attention and (l + 1) % self.neighbor_layer_wise == 0:
                self.neighbor_layers.append(MPTDecoderLayer(config, cross_attention=True))
        self.gradient_checkpointing = False
        self.post_init()

    def get_input_embeddings(self):
        return self.embed_tokens

    def set_input_embeddings(self, value):
        self.embed_tokens = value

    def _prepare_decoder_attention_mask(self, attention_mask, input_shape, inputs_embeds, past_key_values_length):
        combined = None
        if input_shape[-1] > 1:
            combined = _make_causal_mask(input_shape, inputs_embeds.dtype, inputs_embeds.device, past_key_values_length)
        if attention_mask is not None:
            expanded = _expand_mask(attention_mask, inputs_embeds.dtype, tgt_len=input_shape[-1])
            combined = expanded if combined is None else expanded + combined
        return combined

    def forward(self, input_ids=None, attention_mask=None, head_mask=None, past_key_values=None, inputs_embeds=None,
                neighbor_embeds=None, neighbor_attention_mask=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, first_key_valid=False):
        output_attentions = bool(output_attentions)
        output_hidden_states = bool(output_hidden_states)
        return_dict = True if return_dict is None else return_dict
        if use_cache:
            raise ValueError("KV-cache decoding is not implemented (the reference's cross-attention ignores the cache, :275)")
        # past_key_values: a FIXED per-layer key/value prefix (peft prefix tuning, reference model/modelling_self_attention.py:88-93),
        # either peft's prefix-encoder table [P, 2 * n_layers * d] (layer i: keys = columns [2i d, (2i+1) d), values the next d)
        # or a sequence of (key [P or B x P, d], value) pairs, one per layer
        prefix_len = 0
        if past_key_values is not None:
            if torch.is_tensor(past_key_values):
                d_model = self.config.hidden_size
                if past_key_values.dim() != 2 or past_key_values.shape[1] != 2 * len(self.layers) * d_model:
                    raise ValueError(f"past_key_values: expected a prefix table [P, 2 * {len(self.layers)} * {d_model}], got {tuple(past_key_values.shape)}")
                past_key_values = [(past_key_values[:, 2 * i * d_model:(2 * i + 1) * d_model],
                                    past_key_values[:, (2 * i + 1) * d_model:(2 * i + 2) * d_model]) for i in range(len(self.layers))]
            if len(past_key_values) != len(self.layers):
                raise ValueError(f"past_key_values: {len(past_key_values)} entries for {len(self.layers)} layers")
            prefix_len = past_key_values[0][0].shape[-2]
            if self.cross_attention:
                raise ValueError("a key/value prefix is only defined for the plain (self-attention) decoder")
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both decoder_input_ids and decoder_inputs_embeds at the same time")
        elif input_ids is not None:
            input_shape = input_ids.size()
            input_ids = input_ids.view(-1, input_shape[-1])
        elif inputs_embeds is not None:
            input_shape = inputs_embeds.size()[:-1]
        else:
            raise ValueError("You have to specify either decoder_input_ids or decoder_inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        batch_size, seq_length = input_shape
        if attention_mask is None:
            attention_mask = torch.ones(batch_size, seq_length + prefix_len, device=inputs_embeds.device)
        elif attention_mask.shape[1] == seq_length and prefix_len:
            attention_mask = torch.cat([attention_mask.new_ones(batch_size, prefix_len), attention_mask], dim=1)   # prefix keys are always visible
        elif attention_mask.shape[1] != seq_length + prefix_len:
            raise ValueError(f"The provided attention mask has length {attention_mask.shape[1]}, but its length should be "
                             f"{seq_length + prefix_len} (sum of the lengths of current and past inputs)")
        # (output_attentions / head_mask / attention_dropout > 0: the layers route their attention to the general HIP core,
        # ops.attn_general -- reference :237-256; the fused kernels keep every other call)
        # The [B,1,T,T] additive mask (:455-476) is never built: the flash kernels take the [B,T] key mask, causality implied.
        # That equals the reference's finfo.min arithmetic as long as no query row is fully masked, i.e. key 0 of every sample
        # is valid -- true for the right-padded sequences of wikiweb2m/data.py:321-333.  The collate can vouch for it on the
        # host (`first_key_valid`); otherwise the check is a device-side assert: no host synchronisation either way.
        if not first_key_valid:
            torch._assert_async((attention_mask[:, 0] != 0).all(),
                                "MPTDecoder: attention_mask[:, 0] must be 1 for every sample (right-padded sequences)")
        causal_attention_mask = (attention_mask != 0).to(torch.uint8).contiguous()
        key_valid = None
        if neighbor_attention_mask is not None:
            key_valid = _key_valid_from(neighbor_attention_mask).to(torch.uint8).contiguous()
        if neighbor_embeds is not None and neighbor_embeds.dtype != inputs_embeds.dtype:
            neighbor_embeds = neighbor_embeds.to(inputs_embeds.dtype)

        pos_embeds = self.embed_positions(attention_mask, prefix_len)      # positions count the prefix (HF: past_key_values_length)
        if self.project_in is not None:
            inputs_embeds = _lin(self.project_in, inputs_embeds)
        hidden_states = inputs_embeds + pos_embeds

        all_hidden_states = () if output_hidden_states else None
        all_self_attns = () if output_attentions else None
        if head_mask is not None and head_mask.size()[0] != len(self.layers):
            raise ValueError(f"The `head_mask` should be specified for {len(self.layers)} layers, but it is for"
                             f" {head_mask.size()[0]}.")

        defer = not output_hidden_states
        for idx, decoder_layer in enumerate(self.layers):
            if output_hidden_states:
                all_hidden_states += (hidden_states,)
            if self.training and self.layerdrop > 0:
                if torch.rand([]) < self.layerdrop:       # LayerDrop (:581-584); OPT's layerdrop is 0
                    continue
            lhm = head_mask[idx] if head_mask is not None else None
            layer_outputs = decoder_layer(hidden_states, attention_mask=causal_attention_mask, layer_head_mask=lhm,
                                          output_attentions=output_attentions, defer_residual=defer,
                                          past_key_value=None if past_key_values is None else past_key_values[idx])
            if self.cross_attention and neighbor_embeds is not None and (idx + 1) % self.neighbor_layer_wise == 0:
                hidden_states = _materialize(layer_outputs[0])
                neighbor_idx = (idx + 1) // self.neighbor_layer_wise - 1
                layer_outputs = self.neighbor_layers[neighbor_idx](
                    hidden_states, attention_mask=causal_attention_mask, neighbor_embeds=neighbor_embeds,
                    neighbor_attention_mask=key_valid, layer_head_mask=lhm, output_attentions=output_attentions)     # (:613-623)
            hidden_states = layer_outputs[0]
            if output_attentions:
                all_self_attns += (layer_outputs[1],)

        if self.final_layer_norm is not None:
            fln = self.final_layer_norm
            if isinstance(hidden_states, _Deferred):
                _, hidden_states = ops.add_layer_norm_pair(hidden_states.branch, hidden_states.residual, fln.weight, fln.bias, fln.eps,
                                                           hidden_states.p_drop, hidden_states.training)
            else:
                hidden_states = ops.layer_norm(hidden_states, fln.weight, fln.bias, fln.eps)
        hidden_states = _materialize(hidden_states)
        if self.project_out is not None:
            hidden_states = _lin(self.project_out, hidden_states)
        if output_hidden_states:
            all_hidden_states += (hidden_states,)
        if not return_dict:
            return tuple(v for v in [hidden_states, None, all_hidden_states, all_self_attns] if v is not None)
        return BaseModelOutputWithPast(last_hidden_state=hidden_states, past_key_values=None,
                                       hidden_states=all_hidden_states, attentions=all_self_attns)


class MPTModel(MPTPreTrainedModel):
    def __init__(self, config: MPTConfig):
        super().__init__(config)
        self.decoder = MPTDecoder(config)

    def get_input_embeddings(self):
        return self.decoder.embed_tokens

    def set_input_embeddings(self, value):
        self.decoder.embed_tokens = value

    def get_decoder(self):
        return self.decoder

    def forward(self, *args, **kwargs):
        return self.decoder(*args, **kwargs)


def reset_peft_parameters(model):
    """Kept for API parity (reference :719-729); no module of the fork carries lora_A/lora_B/adapter names."""
    for n, p in model.named_parameters():
        if "lora_A" in n:
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))
        if "lora_B" in n:
            nn.init.zeros_(p)


def mark_only_peft_as_trainable(model):
    """Freeze everything, then un-freeze the cross-attention layers (reference :731-737) = the DDP gradient set."""
    for p in model.parameters():
        p.requires_grad = False
    for m in model.modules():
        if isinstance(m, MPTDecoderLayer) and m.cross_attention:
            for p in m.parameters():
                p.requires_grad = True


def lm_head_loss_and_logits(module, lm_head, hidden, next_labels, return_logits=None, logits_slice=None):
    """(loss, logits) of a causal-LM head (reference :826-836).  Default = the reference's contract: full [B, T, V] logits and a
    separate cross-entropy.  A caller that does not need them (the trainer, the benchmark) says so -- `return_logits=False`, or
    `logits_slice` (a slice along T: the logits of just those positions, without gradient: what the trainer's running summary
    loss reads, run_generation.py:473-480 looks at positions max_input_length..T-1 only) -- and a step with a frozen head (training or
    evaluation; every peft mode of the reference: the head is tied to the frozen embedding) then runs ops.lm_head_cross_entropy: lm_head and
    the token cross-entropy fused, the [B, T, V] logits and their gradient (4.1 GB each at B=64) never built.  The fused kernel
    needs the head's dimensions aligned to its 16-byte lanes; otherwise the unfused pair (which pads) runs."""
    frozen = type(lm_head) is nn.Linear and not lm_head.weight.requires_grad and lm_head.bias is None
    lanes = 16 // hidden.element_size()
    aligned = lm_head.weight.shape[0] % lanes == 0 and lm_head.weight.shape[1] % lanes == 0
    opted_out = return_logits is False or (return_logits is None and logits_slice is not None)
    fuse = next_labels is not None and frozen and hidden.is_cuda and opted_out and aligned
    if fuse:
        loss = ops.lm_head_cross_entropy(hidden, lm_head.weight, next_labels)
        logits = None
        if logits_slice is not None:
            with torch.no_grad():
                logits = _lin(lm_head, hidden.detach()[:, logits_slice].contiguous())
        return loss, logits
    logits = _lin(lm_head, hidden)
    loss = None
    if next_labels is not None:
        loss = ops.cross_entropy(logits.view(-1, logits.shape[-1]), next_labels.view(-1))
    if logits_slice is not None:
        logits = logits[:, logits_slice]
    return loss, logits


class MPTForCausalLM(MPTPreTrainedModel):
    _tied_weights_keys = ["lm_head.weight"]

    def __init__(self, config):
        super().__init__(config)
        self.model = MPTModel(config)
        self.lm_head = nn.Linear(config.word_embed_proj_dim, config.vocab_size, bias=False)
        self.lm_head.apply(self._init_weights)
        # the reference-era transformers ties lm_head to embed_tokens in post_init
        self.lm_head.weight = self.model.decoder.embed_tokens.weight
        if config.peft_type != "none":
            reset_peft_parameters(self.model)
            mark_only_peft_as_trainable(self.model)

    def get_input_embeddings(self):
        return self.model.decoder.embed_tokens

    def set_input_embeddings(self, value):
        self.model.decoder.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def set_decoder(self, decoder):
        self.model.decoder = decoder

    def get_decoder(self):
        return self.model.decoder

    def forward(self, input_ids=None, attention_mask=None, head_mask=None, past_key_values=None, inputs_embeds=None,
                labels=None, neighbor_embeds=None, neighbor_attention_mask=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, first_key_valid=False, return_logits=None, logits_slice=None):
        """return_logits / logits_slice: see lm_head_loss_and_logits (training steps never build the [B, T, V] logits unless asked)."""
        return_dict = True if return_dict is None else return_dict
        outputs = self.model.decoder(input_ids=input_ids, attention_mask=attention_mask, head_mask=head_mask,
                                     past_key_values=past_key_values, inputs_embeds=inputs_embeds,
                                     neighbor_embeds=neighbor_embeds, neighbor_attention_mask=neighbor_attention_mask,
                                     use_cache=use_cache, output_attentions=output_attentions,
                                     output_hidden_states=output_hidden_states, return_dict=True, first_key_valid=first_key_valid)
        hidden = outputs.last_hidden_state
        nxt = None
        if labels is not None:
            # tokens < n predict n (:831-836).  Instead of copying the [B,T-1,V] slice, every row is scored against the
            # next label and the last position of each sample is ignored: the same mean over B*(T-1) rows.
            labels = labels.to(hidden.device)
            nxt = torch.full_like(labels, -100)
            nxt[:, :-1] = labels[:, 1:]
        loss, logits = lm_head_loss_and_logits(self, self.lm_head, hidden, nxt, return_logits, logits_slice)
        if not return_dict:
            output = (logits,) + tuple(v for v in (outputs.hidden_states, outputs.attentions) if v is not None)
            return (loss,) + output if loss is not None else output
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=None, hidden_states=outputs.hidden_states,
                                      attentions=outputs.attentions)


def copy_opt_weights(opt_model, mpt_model):
    """HF OPTForCausalLM -> the fork, module by module (reference :959-974); the fork's extra cross-attention layers keep
    their fresh initialisation."""
    src, dst = opt_model.model.decoder, mpt_model.model.decoder
    dst.embed_tokens.load_state_dict(src.embed_tokens.state_dict())
    dst.embed_positions.load_state_dict(src.embed_positions.state_dict())
    if dst.project_in is not None:
        dst.project_out.load_state_dict(src.project_out.state_dict())
        dst.project_in.load_state_dict(src.project_in.state_dict())
    if dst.final_layer_norm is not None:
        dst.final_layer_norm.load_state_dict(src.final_layer_norm.state_dict())
    for idx in range(len(src.layers)):
        missing, unexpected = dst.layers[idx].load_state_dict(src.layers[idx].state_dict(), strict=False)
        if missing or unexpected:
            print(f"{idx}th layer missing_keys: {missing}, unexpected_keys: {unexpected}")
    mpt_model.lm_head.load_state_dict(opt_model.lm_head.state_dict())


class _PatchEmbedLinear(nn.Conv2d):
    """CLIP's patch embedding is a stride == kernel Conv2d, i.e. a GEMM over flattened patches.  MIOpen has no tuned bf16
    kernel for it here and falls back to `naive_conv_*` (8 ms / step at B=8, profiles/r1_b); the GEMM form is the same
    arithmetic.  Parameters and state-dict keys are untouched (the instance's class is swapped in place)."""

    def forward(self, x):
        p = self.kernel_size[0]
        if self.kernel_size != self.stride or self.kernel_size[0] != self.kernel_size[1] or self.padding != (0, 0) or x.shape[-1] % p or x.shape[-2] % p:
            return super().forward(x)
        n, c, hh, ww = x.shape
        cols = x.view(n, c, hh // p, p, ww // p, p).permute(0, 2, 4, 1, 3, 5).reshape(n * (hh // p) * (ww // p), c * p * p)
        w = self.weight.view(self.out_channels, -1)
        if cols.is_cuda:
            # frozen encoder: forward only, on the HIP GEMM.  A caller that wants gradients through the patch embedding on the GPU
            # gets an error, not a silent library GEMM (the reference freezes the visual model: modelling_cross_attention.py:934-939)
            if torch.is_grad_enabled() and (cols.requires_grad or w.requires_grad or (self.bias is not None and self.bias.requires_grad)):
                raise ValueError("_PatchEmbedLinear: the patch embedding of the frozen visual encoder has no HIP backward; run it under "
                                 "torch.no_grad() or freeze it (requires_grad_(False))")
            y = ops.gemm_nt(cols.contiguous(), w, self.bias)
        else:
            y = F.linear(cols, w, self.bias)                        # CPU tensors only (the reference's CPU plumbing path, BASELINE config 1)
        return y.view(n, hh // p, ww // p, self.out_channels).permute(0, 3, 1, 2)


def _conv_patch_embed_as_gemm(module: nn.Module):
    for m in module.modules():
        if type(m) is nn.Conv2d and m.kernel_size == m.stride and m.groups == 1:
            m.__class__ = _PatchEmbedLinear


def _valid_rows_host(pos_ids_cpu, sample_has_key=None):
    """Indices (CPU LongTensor) of neighbor slots with pos_id > 0, or None when every slot is valid OR some sample has no valid
    key at all (then padded slots DO matter: a fully-masked sample attends uniformly over all its keys).  `sample_has_key` [B] bool:
    whether each sample has a valid slot in ANY modality (a sample without images still has its page-info text neighbor, reference
    data.py:355-361 -- its padded image slots get probability exactly 0); default: judged on this modality alone."""
    valid = pos_ids_cpu > 0
    has_key = valid.any(dim=1) if sample_has_key is None else sample_has_key
    if bool(valid.all()) or not bool(has_key.all()):
        return None
    return valid.reshape(-1).nonzero().squeeze(1)


def _valid_rows(pos_ids):
    """Device version of _valid_rows_host for callers without host metadata: one host sync (the [B, N] position ids)."""
    rows = _valid_rows_host(pos_ids.cpu())
    return None if rows is None else rows.to(pos_ids.device)


def host_metadata(batch, use_images=True):
    """What the forward pass needs to know on the HOST, computed from a batch that is still in host memory (the collate
    output, before the H2D copy): which neighbor slots are real (reference data.py:444-454 pads with pos_id 0), how long every
    real neighbor text is (the packing of the padding-free encoder), and that every sequence starts with a valid token.
    Passing it as `host_meta=` removes every device->host synchronisation from the training step."""
    meta = {"first_key_valid": bool((batch["attention_mask"][:, 0] != 0).all())}
    npos = batch.get("neighbor_pos_ids")
    ipos = batch.get("neighbor_images_pos_ids") if use_images else None      # context text_only: image slots are not keys
    has_key = None                                     # per sample: a valid neighbor slot in any modality
    for pos in (npos, ipos):
        if pos is not None:
            has_key = (pos.cpu() > 0).any(dim=1) if has_key is None else has_key | (pos.cpu() > 0).any(dim=1)
    if npos is not None and "neighbor_attention_mask" in batch:
        rows = _valid_rows_host(npos.cpu(), has_key)
        am = batch["neighbor_attention_mask"].cpu()
        am = am.reshape(-1, am.shape[-1]) != 0
        if rows is not None:
            am = am.index_select(0, rows)
        meta["text_rows"] = rows
        meta["text_lens"] = am.sum(1).to(torch.int32)
        meta["text_first_valid"] = bool(am[:, 0].all()) if am.numel() else True
    if ipos is not None:
        meta["image_rows"] = _valid_rows_host(ipos.cpu(), has_key)
    return meta


class TextPooler(nn.Module):
    """CLS token -> Linear -> tanh (reference :879-893)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        first = hidden_states[:, 0].contiguous()
        return torch.tanh(ops.linear(first, self.dense.weight, self.dense.bias))


class CrossAttentionModel(nn.Module):
    """Wrapper: frozen neighbor encoders -> pooled, projected neighbor tokens -> LM with gated cross-attention
    (reference :896-1114).  `lm_config` / `text_config` / `visual_config` (HF config objects) build randomly
    initialised models instead of calling from_pretrained -- for synthetic benchmarks and tests, where no
    checkpoint can be downloaded."""

    def __init__(self, args, tokenizer, lm_config=None, text_config=None, visual_config=None):
        super().__init__()
        self.args = args
        self.context = args.context
        self.neighbor_mode = args.neighbor_mode
        self.n_text_tokens = args.n_text_tokens
        self.n_visual_tokens = args.n_visual_tokens
        self.tokenizer = tokenizer
        self.cross_path = uses_cross_attention(args.neighbor_mode, args.peft_type)

        self.initialize_lm(args, lm_config)
        self.input_embeddings = self.lm.get_input_embeddings()

        self.text_model = None
        if self.context != "section_only":
            embedding_dim = self.input_embeddings.embedding_dim * args.n_text_tokens
            if "clip" in args.text_model:
                self.text_model = CLIPTextModel(text_config) if text_config is not None else CLIPTextModel.from_pretrained(args.text_model)
            else:
                self.text_model = (RobertaModel(text_config, add_pooling_layer=False) if text_config is not None
                                   else RobertaModel.from_pretrained(args.text_model))
                self.text_pooler = TextPooler(self.text_model.config)
            self.text_embeddings = nn.Linear(self.text_model.config.hidden_size, embedding_dim)
            self.text_position_embeddings = nn.Embedding(args.max_output_length + 1, embedding_dim)
            self.text_model.eval()
            for p in self.text_model.parameters():
                p.requires_grad = False

        self.visual_model = None
        if self.context in ("section_all", "all"):
            if args.n_text_tokens != args.n_visual_tokens:
                raise ValueError("n_text_tokens must equal n_visual_tokens: the interleaved neighbor layout shares one "
                                 "n_tokens (reference :1083-1098)")
            embedding_dim = self.input_embeddings.embedding_dim * args.n_visual_tokens
            self.visual_model = (CLIPVisionModel(visual_config) if visual_config is not None
                                 else CLIPVisionModel.from_pretrained(args.visual_model))
            self.visual_embeddings = nn.Linear(self.visual_model.config.hidden_size, embedding_dim)
            self.visual_position_embeddings = nn.Embedding(args.max_output_length + 1, embedding_dim)
            self.visual_model.eval()
            for p in self.visual_model.parameters():
                p.requires_grad = False
            _conv_patch_embed_as_gemm(self.visual_model)
        # Padded neighbor slots (pos_id 0) are masked keys: they can influence neither the logits nor any gradient, so
        # the frozen encoders skip them (the reference encodes '' texts and all-zero images, data.py:444-454).
        self.skip_padded_neighbors = getattr(args, "skip_padded_neighbors", True)
        # packed (padding-free) HIP forward of the frozen encoders -- the only forward they have here; the HF modules stay the owners
        # of the weights.  An encoder architecture none of the HIP forwards cover is an error, never a library-GEMM / SDPA forward.
        self._packed_text = PackedTextEncoder(self.text_model) if (self.text_model is not None and PackedTextEncoder.supports(self.text_model)) else None
        self._clip_text = ClipTextEncoder(self.text_model) if (self.text_model is not None and ClipTextEncoder.supports(self.text_model)) else None
        self._packed_visual = PackedVisionEncoder(self.visual_model) if (self.visual_model is not None and PackedVisionEncoder.supports(self.visual_model)) else None
        if self.text_model is not None and (self._clip_text if "clip" in args.text_model else self._packed_text) is None:
            raise ValueError(f"text_model {type(self.text_model).__name__}: no HIP forward for this architecture (RoBERTa/BERT-style "
                             "absolute-position encoders and CLIP's text tower are covered)")
        if self.visual_model is not None and self._packed_visual is None:
            raise ValueError(f"visual_model {type(self.visual_model).__name__}: no HIP forward for this architecture (CLIP ViT is covered)")

        if self.args.freeze_lm:
            print("Freezing the LM.")
            self.lm.eval()
            for p in self.lm.parameters():
                p.requires_grad = False
        else:
            self.lm.train()

    def initialize_lm(self, args, lm_config=None):
        """HF loading API (reference :951-976): OPT weights are copied into the fork layer by layer.  A "llama" model name
        selects the Llama-family variant (frozen HF LlamaForCausalLM + gated cross-attention layers, no reference counterpart)."""
        if "llama" in args.model_name_or_path.lower() or (lm_config is not None and getattr(lm_config, "model_type", "") == "llama"):
            from .modelling_llama_cross_attention import LlamaNeighborLM
            self.lm = LlamaNeighborLM(args, lm_config)
            return
        if lm_config is not None:
            opt_config, opt_model = lm_config, None
        else:
            opt_config = AutoConfig.from_pretrained(args.model_name_or_path)
            opt_model = AutoModelForCausalLM.from_pretrained(args.model_name_or_path, config=opt_config)
        mpt_model = MPTForCausalLM(MPTConfig(args, opt_config))
        if opt_model is not None:
            copy_opt_weights(opt_model, mpt_model)
        self.lm = mpt_model

    # -------------------------------------------------------------------------------- neighbor encoders
    def _project(self, pooled, linear, pos_emb, pos_ids, batch_size, n_tokens):
        embs = ops.linear(pooled.to(linear.weight.dtype).contiguous(), linear.weight, linear.bias)
        if pos_ids is not None:
            embs = embs + pos_emb(pos_ids.reshape(-1))
        return embs.reshape(batch_size, -1, n_tokens, embs.shape[-1] // n_tokens)

    def get_text_embs(self, input_ids, attention_mask, pos_ids=None, host_meta=None):
        """[B,N,L] ids -> [B,N,n_text_tokens,d]  (reference :978-1004)."""
        batch_size, neighbor_num, seq_len = input_ids.shape
        ids, am = input_ids.reshape(-1, seq_len), attention_mask.reshape(-1, seq_len)
        rows, lens_host = None, None
        if self.skip_padded_neighbors and pos_ids is not None:
            if host_meta is not None and "text_rows" in host_meta:
                rows = host_meta["text_rows"]
                rows = None if rows is None else rows.to(ids.device, non_blocking=True)
                lens_host = (host_meta["text_lens"], host_meta["text_first_valid"])
            else:
                rows = _valid_rows(pos_ids)
        elif host_meta is not None and host_meta.get("text_rows", 0) is None and "text_lens" in host_meta:
            lens_host = (host_meta["text_lens"], host_meta["text_first_valid"])
        with torch.no_grad():
            if rows is not None:
                ids, am = ids.index_select(0, rows), am.index_select(0, rows)
            if "clip" in self.args.text_model:
                enc = self._clip_text.pooled(ids, am)
            else:
                enc = self._packed_text.cls(ids, am, lens_host)
            if rows is not None:
                full = enc.new_zeros(batch_size * neighbor_num, enc.shape[-1])
                enc = full.index_copy_(0, rows, enc)
        if "clip" in self.args.text_model:
            pooled = enc
        else:
            pooled = torch.tanh(ops.linear(enc.contiguous(), self.text_pooler.dense.weight, self.text_pooler.dense.bias))
        return self._project(pooled, self.text_embeddings, self.text_position_embeddings, pos_ids, batch_size, self.n_text_tokens)

    def get_visual_embs(self, pixel_values, pos_ids=None, host_meta=None):
        """[B,N,3,H,W] pixels -> [B,N,n_visual_tokens,d]  (reference :1006-1027)."""
        batch_size, neighbor_num, pixel, width, height = pixel_values.shape
        rows = None
        if self.skip_padded_neighbors and pos_ids is not None:
            if host_meta is not None and "image_rows" in host_meta:
                rows = host_meta["image_rows"]
                rows = None if rows is None else rows.to(pixel_values.device, non_blocking=True)
            else:
                rows = _valid_rows(pos_ids)
        with torch.no_grad():
            pv = pixel_values.reshape(-1, pixel, width, height)
            if rows is not None:
                pv = pv.index_select(0, rows)
            hidden = self.visual_model.config.hidden_size
            if pv.shape[0] == 0:
                pooled = pixel_values.new_zeros(0, hidden, dtype=next(self.visual_model.parameters()).dtype)
            else:
                pv = pv.to(next(self.visual_model.parameters()).dtype)
                pooled = self._packed_visual.pooled(pv)
            if rows is not None:
                pooled = pooled.new_zeros(batch_size * neighbor_num, hidden).index_copy_(0, rows, pooled)
        return self._project(pooled, self.visual_embeddings, self.visual_position_embeddings, pos_ids, batch_size, self.n_visual_tokens)

    def train(self, mode=True):
        super().train(mode=mode)
        if self.args.freeze_lm:
            self.lm.eval()
        if self.text_model is not None:
            self.text_model.eval()
        if self.visual_model is not None:
            self.visual_model.eval()
        return self

    def forward(self, input_ids, attention_mask, labels, images=None, image_positions=None, neighbor_input_ids=None,
                neighbor_attention_mask=None, neighbor_pos_ids=None, text_locations=None, neighbor_images=None,
                neighbor_images_pos_ids=None, image_locations=None, host_meta=None, return_logits=None, logits_slice=None):
        """`host_meta` (optional, not in the reference's signature): the dict of `host_metadata(batch)` computed by the collate /
        trainer while the batch was still in host memory; with it the step has no device->host synchronisation.
        `return_logits` / `logits_slice`: see lm_head_loss_and_logits -- a training step does not build the [B, T, V] logits."""
        if self.neighbor_mode == "raw" or self.context == "section_only":
            neighbor_embeds, key_valid = None, None          # sanity path: the plain OPT (:1068-1071)
        elif self.cross_path and self.context == "text_only":
            text = self.get_text_embs(neighbor_input_ids, neighbor_attention_mask, neighbor_pos_ids, host_meta)
            neighbor_embeds, key_valid = ops.neighbor_interleave(
                text, None, torch.arange(text.shape[1], device=text.device).expand(text.shape[0], -1).contiguous(), None,
                neighbor_pos_ids, None)
        elif self.cross_path and self.context in ("section_all", "all"):
            text = self.get_text_embs(neighbor_input_ids, neighbor_attention_mask, neighbor_pos_ids, host_meta)
            visual = self.get_visual_embs(neighbor_images, neighbor_images_pos_ids, host_meta)
            neighbor_embeds, key_valid = ops.neighbor_interleave(text, visual, text_locations, image_locations,
                                                                 neighbor_pos_ids, neighbor_images_pos_ids)
        else:
            raise ValueError(f"Neighbor mode: {self.neighbor_mode} and context: {self.context} are not supported.")
        return self.lm(input_ids=input_ids, attention_mask=attention_mask, labels=labels, neighbor_embeds=neighbor_embeds,
                       neighbor_attention_mask=key_valid, first_key_valid=bool(host_meta and host_meta.get("first_key_valid")),
                       return_logits=return_logits, logits_slice=logits_slice)
