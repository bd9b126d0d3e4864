"""Llama-family variant of the gated neighbor cross-attention LM (BASELINE.json configs[4]: Llama-2-7B decoder,
flamingo, 32 neighbors).  The reference's fork is OPT-specific (learned positions, LayerNorm, ReLU FFN, biases --
model/modelling_cross_attention.py:278-375); this is the same Flamingo-style block in Llama's conventions -- RMSNorm
pre-norm, bias-free projections, SwiGLU FFN, scalar tanh gates initialised at 0 -- inserted after every
`neighbor_layer_wise`-th layer of a frozen LlamaForCausalLM.  The HF object is loaded through the same HF API and owns the
weights (state-dict keys `llama.*`); its forward is replaced by the HIP path: RMSNorm kernel, ONE fused q|k|v GEMM
(ping-pong MFMA kernel, D^-1/2 folded into the q rows), rotary embedding in place on that buffer, causal flash attention
reading Q/K/V in place, ONE fused gate|up GEMM + SwiGLU kernel, down projection; dgrads against cached W^T copies.
No reference counterpart exists: parity is UNPINNED vs MMGL; it is pinned (tests/test_llama_gpu.py) to HF Llama itself
when the gates are 0 and to the CPU oracle (oracle/llama_ref.py) otherwise.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers.modeling_outputs import CausalLMOutputWithPast

from .. import ops


class LlamaGatedCrossAttentionLayer(nn.Module):
    def __init__(self, hidden_size, num_heads, intermediate_size, rms_norm_eps=1e-6, dropout=0.0):
        super().__init__()
        if hidden_size % num_heads:
            raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {hidden_size} and `num_heads`: {num_heads}).")
        self.num_heads, self.head_dim, self.eps, self.dropout = num_heads, hidden_size // num_heads, rms_norm_eps, dropout
        self.input_layernorm = nn.Parameter(torch.ones(hidden_size))
        self.post_attention_layernorm = nn.Parameter(torch.ones(hidden_size))
        self.q_proj = nn.Linear(hidden_size, hidden_size, bias=False)
        self.k_proj = nn.Linear(hidden_size, hidden_size, bias=False)
        self.v_proj = nn.Linear(hidden_size, hidden_size, bias=False)
        self.o_proj = nn.Linear(hidden_size, hidden_size, bias=False)
        self.gate_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.up_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.down_proj = nn.Linear(intermediate_size, hidden_size, bias=False)
        self.gating1 = nn.Parameter(torch.tensor(0.0))
        self.gating2 = nn.Parameter(torch.tensor(0.0))

    def forward(self, hidden_states, neighbor_embeds, key_valid):
        h = hidden_states
        x = ops.rms_norm(h, self.input_layernorm, self.eps)
        q = ops.linear(x, self.q_proj.weight, None, out_scale=self.head_dim ** -0.5)
        k = ops.linear(neighbor_embeds, self.k_proj.weight, None)
        v = ops.linear(neighbor_embeds, self.v_proj.weight, None)
        a = ops.linear(ops.xattn_core(q, k, v, key_valid, self.num_heads), self.o_proj.weight, None)
        h = ops.gated_residual(h, a, self.gating1, self.dropout, self.training)
        x = ops.rms_norm(h, self.post_attention_layernorm, self.eps)
        # gate and up are separate trainable parameters (state-dict names of LlamaMLP): their WEIGHTS are concatenated per call (180 MB
        # at 7B dims) so that one GEMM writes the [gate | up] buffer the SwiGLU kernel reads and one dgrad / one wgrad GEMM run backward --
        # concatenating the two activations cost 1.5 GB of traffic per layer and a contiguous copy of each gradient half
        gu = ops.linear(x, torch.cat([self.gate_proj.weight, self.up_proj.weight], dim=0), None)
        m = ops.linear(ops.swiglu(gu), self.down_proj.weight, None)
        return ops.gated_residual(h, m, self.gating2, self.dropout, self.training)


class _FrozenLlamaLayer:
    """Functional forward of one frozen HF LlamaDecoderLayer on the HIP kernels (derived fused weights are cached copies; the
    module's own parameters and state_dict stay as loaded)."""

    def __init__(self, layer, cfg):
        self.layer, self.cfg = layer, cfg
        self.H = cfg.num_attention_heads
        self.D = getattr(cfg, "head_dim", None) or cfg.hidden_size // self.H
        self._cache = None

    def _fused(self):
        at, mlp = self.layer.self_attn, self.layer.mlp
        ps = (at.q_proj.weight, at.k_proj.weight, at.v_proj.weight, mlp.gate_proj.weight, mlp.up_proj.weight)
        key = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
        if self._cache is None or self._cache[0] != key:
            with torch.no_grad():
                qkv = torch.cat([ps[0].float() * self.D ** -0.5, ps[1].float(), ps[2].float()], 0).to(ps[0].dtype).contiguous()
                gu = torch.cat([ps[3], ps[4]], 0).contiguous()
            self._cache = (key, qkv, gu)
        return self._cache[1], self._cache[2]

    def __call__(self, h, pending, key_valid, cos_sin):
        """(h, pending) -> (h', pending'): the residual stream and the MLP output NOT yet added to it -- the add runs inside the
        RMSNorm kernel of whoever consumes the sum next (this layer's successor, the final norm), forward and backward."""
        ly, eps = self.layer, self.cfg.rms_norm_eps
        w_qkv, w_gu = self._fused()
        if pending is None:
            x = ops.rms_norm(h, ly.input_layernorm.weight, eps)
        else:
            h, x = ops.add_rms_norm_pair(pending, h, ly.input_layernorm.weight, eps)
        qkv = ops.rope_qk_(ops.frozen_linear(x, w_qkv, None), cos_sin, self.H)
        a = ops.frozen_linear(ops.selfattn_core_fused(qkv, key_valid, self.H), ly.self_attn.o_proj.weight, None)
        h, x = ops.add_rms_norm_pair(a, h, ly.post_attention_layernorm.weight, eps)
        m = ops.frozen_linear(ops.swiglu(ops.frozen_linear(x, w_gu, None)), ly.mlp.down_proj.weight, None)
        return h, m


class LlamaNeighborLM(nn.Module):
    """Frozen LlamaForCausalLM (HF loading API, weights owned by the HF module) + trainable gated cross-attention layers;
    same call contract as MPTForCausalLM."""

    def __init__(self, args, llama_config=None):
        super().__init__()
        from transformers import AutoConfig, AutoModelForCausalLM, LlamaForCausalLM
        if llama_config is not None:
            self.llama = LlamaForCausalLM(llama_config)
        else:
            cfg = AutoConfig.from_pretrained(args.model_name_or_path)
            self.llama = AutoModelForCausalLM.from_pretrained(args.model_name_or_path, config=cfg)
        cfg = self.llama.config
        self.config = cfg
        if getattr(cfg, "num_key_value_heads", cfg.num_attention_heads) != cfg.num_attention_heads:
            raise ValueError("LlamaNeighborLM: grouped-query attention (num_key_value_heads != num_attention_heads) is not implemented "
                             "(Llama-2-7B, the BASELINE config, is multi-head)")
        if getattr(cfg, "attention_bias", False) or getattr(cfg, "mlp_bias", False):
            raise ValueError("LlamaNeighborLM: biased projections are not implemented")
        for p in self.llama.parameters():
            p.requires_grad = False
        n_layers = cfg.num_hidden_layers
        wise = getattr(args, "neighbor_layer_wise", None) or max(1, n_layers // max(1, int(getattr(args, "num_neighbor_layers", 4))))
        self.neighbor_layer_wise = int(wise)
        self.neighbor_layers = nn.ModuleList(
            LlamaGatedCrossAttentionLayer(cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size, cfg.rms_norm_eps)
            for l in range(n_layers) if (l + 1) % self.neighbor_layer_wise == 0)
        std = getattr(cfg, "initializer_range", 0.02)
        for m in self.neighbor_layers.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=std)
        self._frozen = [_FrozenLlamaLayer(layer, cfg) for layer in self.llama.model.layers]
        self._rope = None
        # The rotary frequencies, kept OUT of the module's buffers: `model.bfloat16()` / `.to(torch.bfloat16)` (what run_generation.py
        # does to the whole model, reference :304-307) casts HF's non-persistent `inv_freq` buffer as well, and positions x
        # frequencies rounded to 8 bits are radians off at T = 2176 (a 4-layer model then drifts 6 % per layer from its fp32 self:
        # tests/test_full_size_gpu.py found it).  A plain attribute is not touched by dtype casts.
        self._inv_freq = self.llama.model.rotary_emb.inv_freq.detach().to(torch.float32).clone()

    def get_input_embeddings(self):
        return self.llama.get_input_embeddings()

    def _cos_sin(self, T, device):
        """fp32 [T, D/2, 2] table of (cos, sin) for positions 0..T-1 from the HF rotary module's own inv_freq / scaling."""
        rot = self.llama.model.rotary_emb
        key = (T, device)
        if self._rope is None or self._rope[0] != key:
            inv = self._inv_freq.to(device=device)
            ang = torch.arange(T, device=device, dtype=torch.float32)[:, None] * inv[None, :]
            sc = float(getattr(rot, "attention_scaling", 1.0))
            self._rope = (key, torch.stack([ang.cos() * sc, ang.sin() * sc], dim=-1).contiguous())
        return self._rope[1]

    def forward(self, input_ids=None, attention_mask=None, labels=None, neighbor_embeds=None, neighbor_attention_mask=None,
                first_key_valid=False, return_logits=None, logits_slice=None, **kw):
        emb = self.llama.get_input_embeddings()
        h = emb(input_ids)
        B, T = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones(B, T, dtype=torch.long, device=h.device)
        if not first_key_valid:                   # same precondition (and device-side check) as MPTDecoder.forward
            torch._assert_async((attention_mask[:, 0] != 0).all(), "LlamaNeighborLM: attention_mask[:, 0] must be 1 for every sample")
        key_mask = (attention_mask != 0).to(torch.uint8).contiguous()
        ne = valid = None
        if neighbor_embeds is not None:
            valid = neighbor_attention_mask
            if valid is None:
                valid = torch.ones(neighbor_embeds.shape[:2], dtype=torch.uint8, device=neighbor_embeds.device)
            ne, valid = neighbor_embeds.to(emb.weight.dtype), valid.to(torch.uint8).contiguous()
        cos_sin = self._cos_sin(T, h.device)
        k = 0
        pending = None                            # a frozen layer's MLP output, added inside the next RMSNorm kernel
        for l, layer in enumerate(self._frozen):
            h, pending = layer(h, pending, key_mask, cos_sin)
            if (l + 1) % self.neighbor_layer_wise == 0:
                if ne is not None:
                    h, pending = ops.gated_residual(h, pending), None
                    h = self.neighbor_layers[k](h, ne, valid)
                k += 1
        if pending is None:
            hidden = ops.rms_norm(h, self.llama.model.norm.weight, self.config.rms_norm_eps)
        else:
            hidden = ops.add_rms_norm_pair(pending, h, self.llama.model.norm.weight, self.config.rms_norm_eps)[1]
        nxt = None
        if labels is not None:
            nxt = torch.full_like(labels, -100)
            nxt[:, :-1] = labels[:, 1:]
        from .modelling_cross_attention import lm_head_loss_and_logits
        loss, logits = lm_head_loss_and_logits(self, self.llama.lm_head, hidden, nxt, return_logits, logits_slice)
        return CausalLMOutputWithPast(loss=loss, logits=logits)
