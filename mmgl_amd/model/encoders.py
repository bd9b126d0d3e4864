"""Padding-free forward of the frozen neighbor encoders.

The reference pushes every neighbor through a frozen HF encoder at full padded length and keeps one vector of it
(`get_text_embs` / `get_visual_embs`, reference model/modelling_cross_attention.py:978-1027: RoBERTa CLS hidden state,
CLIP-ViT `pooler_output`).  Here the SAME HF modules are loaded through the same `from_pretrained` API and keep their
parameters / state_dict keys; only their forward is replaced, for CUDA tensors, by a packed pass:

  * valid tokens of all neighbor sequences are concatenated ([ntok, hidden], `cu_seqlens`); no pad token is embedded,
    multiplied or attended to (dropping masked keys leaves every softmax unchanged: their weight is exactly 0);
  * per layer: one fused-QKV GEMM (`mmgl_gemm_nt`: the persistent ping-pong MFMA kernel, the D^-1/2 scaling folded into
    the Q rows), `mmgl_encattn_fwd` (hand-written flash-style HIP kernel reading Q/K/V straight out of the fused buffer),
    output GEMM, `mmgl_add_layernorm_fwd` (residual add + LayerNorm in one pass), FFN GEMMs with GELU / quick-GELU in the
    epilogue of the first one;
  * the last layer only produces what is consumed: keys/values for all tokens, but attention output, output projection,
    FFN and norms for the first (CLS) row of every sequence only.

CLIP's TEXT tower (reference :918-921 accepts a `CLIPTextModel` as text_model) is a causal pre-LN encoder whose pooled
output is the hidden state of the EOS token: `ClipTextEncoder` runs it on the same kernels (fused-QKV GEMM, the causal
self-attention kernel of the LM layers with the attention mask as key mask, add+LayerNorm, quick-GELU epilogue) at its
padded length -- CLIP texts are at most 77 tokens.

Forward only, no autograd (the encoders are frozen, reference :922-934).  An architecture none of these cover
(`supports()` is False: relative position embeddings, exotic activations) is an error at construction time -- the product has
no HuggingFace / library-GEMM forward to fall back to (HF forwards appear in tests only, as the thing compared against);
inputs the kernels cannot take (CPU tensors, a sequence whose first token is masked) raise.
"""
import math

import torch

from .. import ops

_HEAD_DIMS = (16, 32, 64, 128)


def _key(*params):
    return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in params)


class _Fused:
    """Per-layer fused QKV weight/bias (Q rows pre-scaled), rebuilt when the source parameters change."""

    def __init__(self):
        self.key = None
        self.layers = None

    def get(self, triples, scale):
        key = _key(*[m.weight for t in triples for m in t], *[m.bias for t in triples for m in t])
        if key != self.key:
            layers = []
            for q, k, v in triples:
                w = torch.cat([q.weight.detach().float() * scale, k.weight.detach().float(), v.weight.detach().float()], 0).to(q.weight.dtype)
                b = torch.cat([q.bias.detach().float() * scale, k.bias.detach().float(), v.bias.detach().float()], 0).to(q.weight.dtype)
                layers.append((w.contiguous(), b.contiguous()))
            self.key, self.layers = key, layers
        return self.layers


def _cu_from_lens(lens):
    cu = torch.zeros(lens.numel() + 1, dtype=torch.int32, device=lens.device)
    cu[1:] = torch.cumsum(lens, 0)
    return cu


class PackedTextEncoder:
    """RoBERTa/BERT-style post-LN encoder (`RobertaModel`): CLS hidden state of the last layer for [n, L] id / mask rows."""

    def __init__(self, model):
        self.model = model
        self._fused = _Fused()
        self._pos_type = (None, None)

    @staticmethod
    def supports(model):
        cfg = getattr(model, "config", None)
        emb = getattr(model, "embeddings", None)
        enc = getattr(model, "encoder", None)
        if cfg is None or emb is None or enc is None or not hasattr(enc, "layer"):
            return False
        if not all(hasattr(emb, a) for a in ("word_embeddings", "position_embeddings", "token_type_embeddings", "LayerNorm")):
            return False
        if getattr(cfg, "position_embedding_type", None) not in (None, "absolute") or getattr(cfg, "is_decoder", False):
            return False
        if not isinstance(cfg.hidden_act, str) or cfg.hidden_act not in ops.ACT_CODES:
            return False
        D = cfg.hidden_size // cfg.num_attention_heads
        return D in _HEAD_DIMS and cfg.hidden_size % 8 == 0 and cfg.intermediate_size % 8 == 0

    def _pos_type_table(self):
        emb = self.model.embeddings
        key = _key(emb.position_embeddings.weight, emb.token_type_embeddings.weight)
        if self._pos_type[0] != key:        # token_type_ids default to 0 everywhere (HF buffer of zeros)
            w = emb.position_embeddings.weight
            self._pos_type = (key, (w.detach().float() + emb.token_type_embeddings.weight.detach()[0].float()).to(w.dtype))
        return self._pos_type[1]

    @torch.no_grad()
    def cls(self, ids, am, lens_host=None):
        """lens_host = (int32 CPU tensor of the row lengths, bool: every row starts with a valid token), known on the host
        from the collate (modelling_cross_attention.host_metadata): with it this pass never synchronises with the device."""
        m = self.model
        cfg = m.config
        if not ids.is_cuda:
            raise RuntimeError("PackedTextEncoder: GPU tensors only (mmgl_amd has no CPU path)")
        n, L = ids.shape
        if n == 0:
            return ids.new_zeros(0, cfg.hidden_size, dtype=m.embeddings.word_embeddings.weight.dtype)
        amb = am != 0
        lens = amb.sum(1)
        if lens_host is None:
            host = torch.stack([lens, amb[:, 0].to(lens.dtype)]).cpu()      # the one host sync of the text pass
            lens_cpu, first_ok = host[0], bool(host[1].all())
        else:
            lens_cpu, first_ok = lens_host
            if lens_cpu.numel() != n:
                raise ValueError(f"PackedTextEncoder: host metadata describes {lens_cpu.numel()} sequences, the batch has {n}")
        if not first_ok:
            raise ValueError("PackedTextEncoder: every neighbor text must start with a valid token (the CLS row is row 0 of its "
                             "packed sequence); tokenizers pad on the right (reference data.py:457)")
        total, max_len = int(lens_cpu.sum()), int(lens_cpu.max())
        cu = _cu_from_lens(lens)
        flat = amb.reshape(-1)
        tok = torch.nonzero_static(flat, size=total).squeeze(1) if hasattr(torch, "nonzero_static") else flat.nonzero().squeeze(1)

        emb = m.embeddings
        pad = emb.padding_idx if getattr(emb, "padding_idx", None) is not None else cfg.pad_token_id
        ne = ids.ne(pad)
        pos = (torch.cumsum(ne, 1) * ne + pad).reshape(-1)                   # create_position_ids_from_input_ids
        x = emb.word_embeddings.weight.index_select(0, ids.reshape(-1).index_select(0, tok))
        pe = self._pos_type_table().index_select(0, pos.index_select(0, tok))
        h = ops.add_layer_norm(x, pe, emb.LayerNorm.weight, emb.LayerNorm.bias, cfg.layer_norm_eps)

        H = cfg.num_attention_heads
        hid = cfg.hidden_size
        layers = list(m.encoder.layer)
        fused = self._fused.get([(l.attention.self.query, l.attention.self.key, l.attention.self.value) for l in layers],
                                1.0 / math.sqrt(hid // H))
        first_rows = cu[:-1].long()
        act = ops.ACT_CODES[cfg.hidden_act]
        es = h.element_size()                # algorithmic work of one attention call (for the bench's per-kernel report)
        work_all = dict(flops=4.0 * float((lens_cpu.double() ** 2).sum()) * hid, bytes=4.0 * total * hid * es)
        work_cls = dict(flops=4.0 * total * hid, bytes=2.0 * (total + n) * hid * es)
        for li, layer in enumerate(layers):
            last = li == len(layers) - 1
            qkv = ops.gemm_nt(h, *fused[li])
            ctx = ops.encoder_attention(qkv[:, :hid], qkv[:, hid:2 * hid], qkv[:, 2 * hid:], cu, H, max_len, q_rows=1 if last else None,
                                        work=work_cls if last else work_all)
            if last:                         # only the CLS rows are consumed downstream
                ctx, h = ctx.index_select(0, first_rows), h.index_select(0, first_rows)
            ao = layer.attention.output
            h1 = ops.add_layer_norm(ops.gemm_nt(ctx, ao.dense.weight, ao.dense.bias), h, ao.LayerNorm.weight, ao.LayerNorm.bias, cfg.layer_norm_eps)
            f = ops.gemm_nt(h1, layer.intermediate.dense.weight, layer.intermediate.dense.bias, act=act)
            lo = layer.output
            h = ops.add_layer_norm(ops.gemm_nt(f, lo.dense.weight, lo.dense.bias), h1, lo.LayerNorm.weight, lo.LayerNorm.bias, cfg.layer_norm_eps)
        return h                             # [n, hidden]


class PackedVisionEncoder:
    """CLIP ViT (pre-LN) encoder: `pooler_output` = post_layernorm(CLS of the last layer) for [n, 3, H, W] pixels."""

    def __init__(self, model):
        self.model = model
        self._fused = _Fused()

    @staticmethod
    def _core(model):
        return getattr(model, "vision_model", model)

    @staticmethod
    def supports(model):
        core = PackedVisionEncoder._core(model)
        cfg = getattr(model, "config", None)
        if cfg is None or not all(hasattr(core, a) for a in ("embeddings", "pre_layrnorm", "encoder", "post_layernorm")):
            return False
        if not hasattr(core.encoder, "layers") or not isinstance(cfg.hidden_act, str) or cfg.hidden_act not in ops.ACT_CODES:
            return False
        D = cfg.hidden_size // cfg.num_attention_heads
        return D in _HEAD_DIMS and cfg.hidden_size % 8 == 0 and cfg.intermediate_size % 8 == 0

    @torch.no_grad()
    def pooled(self, pixel_values):
        core = self._core(self.model)
        cfg = self.model.config
        if not pixel_values.is_cuda:
            raise RuntimeError("PackedVisionEncoder: GPU tensors only (mmgl_amd has no CPU path)")
        if pixel_values.shape[0] == 0:
            return pixel_values.new_zeros(0, cfg.hidden_size)
        e = core.embeddings(pixel_values)                                   # patch GEMM + class token + positions: [n, S, hid]
        n, S, hid = e.shape
        H = cfg.num_attention_heads
        eps = cfg.layer_norm_eps
        x = ops.layer_norm(e.reshape(n * S, hid), core.pre_layrnorm.weight, core.pre_layrnorm.bias, eps)
        cu = torch.arange(0, (n + 1) * S, S, dtype=torch.int32, device=x.device)
        layers = list(core.encoder.layers)
        fused = self._fused.get([(l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj) for l in layers], 1.0 / math.sqrt(hid // H))
        first_rows = cu[:-1].long()
        act = ops.ACT_CODES[cfg.hidden_act]
        y = ops.layer_norm(x, layers[0].layer_norm1.weight, layers[0].layer_norm1.bias, eps) if layers else None
        es = x.element_size()
        work_all = dict(flops=4.0 * n * S * S * hid, bytes=4.0 * n * S * hid * es)
        work_cls = dict(flops=4.0 * n * S * hid, bytes=2.0 * (n * S + n) * hid * es)
        for li, layer in enumerate(layers):
            last = li == len(layers) - 1
            qkv = ops.gemm_nt(y, *fused[li])
            ctx = ops.encoder_attention(qkv[:, :hid], qkv[:, hid:2 * hid], qkv[:, 2 * hid:], cu, H, S, q_rows=1 if last else None,
                                        work=work_cls if last else work_all)
            if last:
                ctx, x = ctx.index_select(0, first_rows), x.index_select(0, first_rows)
            a = ops.gemm_nt(ctx, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)
            x, y2 = ops.add_layer_norm(a, x, layer.layer_norm2.weight, layer.layer_norm2.bias, eps, return_sum=True)
            f = ops.gemm_nt(y2, layer.mlp.fc1.weight, layer.mlp.fc1.bias, act=act)
            f2 = ops.gemm_nt(f, layer.mlp.fc2.weight, layer.mlp.fc2.bias)
            if last:
                return ops.add_layer_norm(f2, x, core.post_layernorm.weight, core.post_layernorm.bias, eps)
            nxt = layers[li + 1].layer_norm1
            x, y = ops.add_layer_norm(f2, x, nxt.weight, nxt.bias, eps, return_sum=True)
        return ops.layer_norm(x.index_select(0, first_rows), core.post_layernorm.weight, core.post_layernorm.bias, eps)


class ClipTextEncoder:
    """CLIP text tower (`CLIPTextModel`): `pooler_output` = final_layer_norm(hidden)[EOS position] for [n, L] id / mask rows
    (transformers CLIPTextTransformer.forward).  Causal attention + right padding: a valid token never sees a pad token, so the
    valid rows equal HF's; pad rows are computed (L <= 77) and never read."""

    def __init__(self, model):
        self.model = model
        self._fused = _Fused()

    @staticmethod
    def _core(model):
        return getattr(model, "text_model", model)

    @staticmethod
    def supports(model):
        core = ClipTextEncoder._core(model)
        cfg = getattr(model, "config", None)
        if cfg is None or not all(hasattr(core, a) for a in ("embeddings", "encoder", "final_layer_norm")):
            return False
        emb = core.embeddings
        if not all(hasattr(emb, a) for a in ("token_embedding", "position_embedding")) or not hasattr(core.encoder, "layers"):
            return False
        if not isinstance(cfg.hidden_act, str) or cfg.hidden_act not in ops.ACT_CODES:
            return False
        D = cfg.hidden_size // cfg.num_attention_heads
        return D in _HEAD_DIMS and cfg.hidden_size % 8 == 0 and cfg.intermediate_size % 8 == 0

    @torch.no_grad()
    def pooled(self, ids, am):
        core = self._core(self.model)
        cfg = self.model.config
        if not ids.is_cuda:
            raise RuntimeError("ClipTextEncoder: GPU tensors only (mmgl_amd has no CPU path)")
        n, L = ids.shape
        hid, H, eps = cfg.hidden_size, cfg.num_attention_heads, cfg.layer_norm_eps
        emb = core.embeddings
        if n == 0:
            return ids.new_zeros(0, hid, dtype=emb.token_embedding.weight.dtype)
        x = (emb.token_embedding.weight.index_select(0, ids.reshape(-1)).view(n, L, hid) + emb.position_embedding.weight[:L]).reshape(n * L, hid)
        layers = list(core.encoder.layers)
        fused = self._fused.get([(l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj) for l in layers], 1.0 / math.sqrt(hid // H))
        act = ops.ACT_CODES[cfg.hidden_act]
        amc = am.contiguous()
        y = ops.layer_norm(x, layers[0].layer_norm1.weight, layers[0].layer_norm1.bias, eps) if layers else None
        for li, layer in enumerate(layers):
            qkv = ops.gemm_nt(y, *fused[li])
            ctx = ops.selfattn_core_fused(qkv.view(n, L, 3 * hid), amc, H).reshape(n * L, hid)
            a = ops.gemm_nt(ctx, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)
            x, y2 = ops.add_layer_norm(a, x, layer.layer_norm2.weight, layer.layer_norm2.bias, eps, return_sum=True)
            f = ops.gemm_nt(y2, layer.mlp.fc1.weight, layer.mlp.fc1.bias, act=act)
            f2 = ops.gemm_nt(f, layer.mlp.fc2.weight, layer.mlp.fc2.bias)
            nxt = layers[li + 1].layer_norm1 if li + 1 < len(layers) else core.final_layer_norm
            x, y = ops.add_layer_norm(f2, x, nxt.weight, nxt.bias, eps, return_sum=True)
        h = y if layers else ops.layer_norm(x, core.final_layer_norm.weight, core.final_layer_norm.bias, eps)
        # the EOS position: transformers takes argmax(ids) for the legacy eos_token_id == 2 configs (the EOS id is the largest of
        # CLIP's vocabulary), else the first position holding eos_token_id
        eos_id = getattr(cfg, "eos_token_id", 2)
        if eos_id == 2:
            pos = ids.argmax(dim=-1)
        else:
            pos = (ids == eos_id).int().argmax(dim=-1)
        return h.view(n, L, hid)[torch.arange(n, device=ids.device), pos]
