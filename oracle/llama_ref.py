"""Oracle (TEST INFRASTRUCTURE): fp32 CPU restatement of the Llama-convention gated cross-attention block
(RMSNorm pre-norm, bias-free projections, SwiGLU, tanh gates) used by mmgl_amd.model.modelling_llama_cross_attention.
There is NO reference counterpart (MMGL's fork is OPT-only): this oracle defines the block from the Flamingo recipe the
reference follows for OPT (model/modelling_cross_attention.py:304-375) with Llama's layer conventions; parity vs MMGL is
unpinned, the attention core inside it is the pinned one (oracle/lm_ref.py attention_core)."""
import torch
import torch.nn.functional as F

from .lm_ref import attention_core, expand_mask


def rms_norm(x, weight, eps):
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)) * weight


def gated_block(p, pre, h, neighbor_embeds, key_valid, num_heads, eps):
    D = h.shape[-1] // num_heads
    x = rms_norm(h, p[pre + "input_layernorm"], eps)
    q = F.linear(x, p[pre + "q_proj.weight"]) * (D ** -0.5)
    k = F.linear(neighbor_embeds, p[pre + "k_proj.weight"])
    v = F.linear(neighbor_embeds, p[pre + "v_proj.weight"])
    a = attention_core(q, k, v, expand_mask(key_valid, h.dtype, h.shape[1]), num_heads)
    h = h + torch.tanh(p[pre + "gating1"]) * F.linear(a, p[pre + "o_proj.weight"])
    x = rms_norm(h, p[pre + "post_attention_layernorm"], eps)
    m = F.silu(F.linear(x, p[pre + "gate_proj.weight"])) * F.linear(x, p[pre + "up_proj.weight"])
    return h + torch.tanh(p[pre + "gating2"]) * F.linear(m, p[pre + "down_proj.weight"])


def llama_neighbor_lm_forward(hf_llama, p, wise, input_ids, attention_mask, labels, neighbor_embeds, key_valid):
    """Run a (CPU, fp32) HF LlamaForCausalLM with the oracle block applied after every `wise`-th layer."""
    cfg = hf_llama.config
    handles, k = [], 0
    for l, layer in enumerate(hf_llama.model.layers):
        if (l + 1) % wise == 0:
            def hook(mod, inp, out, k=k):
                hs = out if torch.is_tensor(out) else out[0]
                hs = gated_block(p, f"neighbor_layers.{k}.", hs, neighbor_embeds, key_valid, cfg.num_attention_heads, cfg.rms_norm_eps)
                return hs if torch.is_tensor(out) else (hs,) + tuple(out[1:])
            handles.append(layer.register_forward_hook(hook))
            k += 1
    try:
        hidden = hf_llama.model(input_ids=input_ids, attention_mask=attention_mask, use_cache=False).last_hidden_state
    finally:
        for h_ in handles:
            h_.remove()
    logits = hf_llama.lm_head(hidden)
    loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1))
    return logits, loss
