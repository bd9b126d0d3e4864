"""Llama-family variant of the gated neighbor cross-attention LM (BASELINE.json configs[4]: Llama-2-7B decoder,
flamingo, 32 neighbors).  The reference's fork is OPT-specific (learned positions, LayerNorm, ReLU FFN, biases --
model/modelling_cross_attention.py:278-375); this is the same Flamingo-style block in Llama's conventions -- RMSNorm
pre-norm, bias-free projections, SwiGLU FFN, scalar tanh gates initialised at 0 -- inserted after every
`neighbor_layer_wise`-th layer of a frozen HuggingFace LlamaForCausalLM through forward hooks, so nothing depends on the
internals of the transformers implementation (RoPE, cache, masks stay HF's).  No reference counterpart exists: parity is
UNPINNED vs MMGL; it is pinned (tests/test_llama_gpu.py) to HF Llama itself when the gates are 0 and to the CPU oracle
(oracle/llama_ref.py) otherwise.  All trainable ops run on the HIP kernels (rms_norm, linear, xattn_core, gated_residual,
cross_entropy).
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
        m = F.silu(ops.linear(x, self.gate_proj.weight, None)) * ops.linear(x, self.up_proj.weight, None)
        m = ops.linear(m, self.down_proj.weight, None)
        return ops.gated_residual(h, m, self.gating2, self.dropout, self.training)


class LlamaNeighborLM(nn.Module):
    """Frozen HF LlamaForCausalLM + trainable gated cross-attention layers; same call contract as MPTForCausalLM."""

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
        self._ctx = None
        k = 0
        for l, layer in enumerate(self.llama.model.layers):
            if (l + 1) % self.neighbor_layer_wise == 0:
                layer.register_forward_hook(self._make_hook(k))
                k += 1

    def _make_hook(self, k):
        def hook(module, inputs, output):
            if self._ctx is None:
                return None
            ne, valid = self._ctx
            if torch.is_tensor(output):
                return self.neighbor_layers[k](output, ne, valid)
            return (self.neighbor_layers[k](output[0], ne, valid),) + tuple(output[1:])
        return hook

    def get_input_embeddings(self):
        return self.llama.get_input_embeddings()

    def forward(self, input_ids=None, attention_mask=None, labels=None, neighbor_embeds=None, neighbor_attention_mask=None, **kw):
        if neighbor_embeds is not None:
            valid = neighbor_attention_mask
            if valid is None:
                valid = torch.ones(neighbor_embeds.shape[:2], dtype=torch.uint8, device=neighbor_embeds.device)
            emb_dtype = self.llama.get_input_embeddings().weight.dtype
            self._ctx = (neighbor_embeds.to(emb_dtype), valid.to(torch.uint8).contiguous())
        try:
            hidden = self.llama.model(input_ids=input_ids, attention_mask=attention_mask, use_cache=False).last_hidden_state
        finally:
            self._ctx = None
        logits = self.llama.lm_head(hidden).contiguous()
        loss = None
        if labels is not None:
            nxt = torch.full_like(labels, -100)
            nxt[:, :-1] = labels[:, 1:]
            loss = ops.cross_entropy(logits.view(-1, logits.shape[-1]), nxt.view(-1))
        return CausalLMOutputWithPast(loss=loss, logits=logits)
