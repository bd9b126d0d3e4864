"""Stage by stage: every op of one frozen Llama layer at 7B dims, bf16 vs fp32 ON THE SAME (bf16-rounded) INPUTS."""
import os, sys, copy
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from helpers import mpt_args
from transformers import LlamaConfig
from mmgl_amd.model.modelling_llama_cross_attention import LlamaNeighborLM
from mmgl_amd import ops

def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20))
cfg = LlamaConfig(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32,
                  max_position_embeddings=4096, pad_token_id=0, bos_token_id=1, eos_token_id=2, attention_dropout=0.0)
torch.manual_seed(5)
lm = LlamaNeighborLM(mpt_args(model_name_or_path="llama-2-7b", neighbor_layer_wise=1), cfg).bfloat16().cuda()
lm32 = copy.deepcopy(lm).float()
B, T = 1, 2176
g = torch.Generator().manual_seed(2)
h = torch.randn(B, T, 4096, generator=g).bfloat16().cuda()
am = torch.ones(B, T, dtype=torch.uint8); am[0, 1400:2048] = 0; am[0, 2100:] = 0
am = am.cuda()
fr, fr32 = lm._frozen[0], lm32._frozen[0]
cs = lm._cos_sin(T, h.device)
eps = cfg.rms_norm_eps
wq, wg = fr._fused(); wq32, wg32 = fr32._fused()
ly, ly32 = fr.layer, fr32.layer
def both(name, f16, f32, *ins):
    a = f16(*ins); b = f32(*[i.float() if i.is_floating_point() else i for i in ins])
    print(f"{name:14s} rel err {rel(a, b):.4f}   (|out| {float(b.float().norm()):.3e})")
    return a
with torch.no_grad():
    x = both("rms_norm", lambda t: ops.rms_norm(t, ly.input_layernorm.weight, eps), lambda t: ops.rms_norm(t, ly32.input_layernorm.weight, eps), h)
    qkv = both("qkv gemm", lambda t: ops.frozen_linear(t, wq, None), lambda t: ops.frozen_linear(t, wq32, None), x)
    qkvr = both("rope", lambda t: ops.rope_qk_(t.clone(), cs, 32), lambda t: ops.rope_qk_(t.clone(), cs, 32), qkv)
    at = both("selfattn", lambda t: ops.selfattn_core_fused(t, am, 32), lambda t: ops.selfattn_core_fused(t, am, 32), qkvr)
    a = both("o_proj", lambda t: ops.frozen_linear(t, ly.self_attn.o_proj.weight, None), lambda t: ops.frozen_linear(t, ly32.self_attn.o_proj.weight, None), at)
    h1 = both("residual", lambda u, v: ops.gated_residual(u, v), lambda u, v: ops.gated_residual(u, v), h, a)
    x2 = both("rms_norm2", lambda t: ops.rms_norm(t, ly.post_attention_layernorm.weight, eps), lambda t: ops.rms_norm(t, ly32.post_attention_layernorm.weight, eps), h1)
    gu = both("gate|up gemm", lambda t: ops.frozen_linear(t, wg, None), lambda t: ops.frozen_linear(t, wg32, None), x2)
    sw = both("swiglu", lambda t: ops.swiglu(t), lambda t: ops.swiglu(t), gu)
    m = both("down_proj", lambda t: ops.frozen_linear(t, ly.mlp.down_proj.weight, None), lambda t: ops.frozen_linear(t, ly32.mlp.down_proj.weight, None), sw)
# backward of the attention alone at this shape
q = qkvr.detach().clone().requires_grad_(); q32 = qkvr.detach().float().requires_grad_()
w = (torch.randn(B, T, 4096, generator=g) * 1e-3).cuda()
ops.selfattn_core_fused(q, am, 32).backward(w.bfloat16()); ops.selfattn_core_fused(q32, am, 32).backward(w.bfloat16().float())
d = 4096
for i, nm in enumerate(("dq", "dk", "dv")):
    print(f"selfattn bwd {nm}: rel err {rel(q.grad[..., i*d:(i+1)*d], q32.grad[..., i*d:(i+1)*d]):.4f}")
