"""torch.autograd.Function wrappers over the C-ABI kernels (one per fwd/bwd pair).

Each op validates shapes the way the reference does (ValueError), makes inputs contiguous, calls the HIP
entry point on torch's current stream and returns fresh tensors owned by autograd.  GPU only.
"""
import math
import os
import threading
import weakref

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import dtype_code, lib, ptr, ptr_off, require_cuda, stream_ptr


# ------------------------------------------------------------------------------------------ attention core
class _XAttnCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_valid, num_heads):
        require_cuda(q, k, v, key_valid)
        B, T, d = q.shape
        S = k.shape[1]
        D = d // num_heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = torch.empty_like(q)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=q.device)
        _lib.call("mmgl_xattn_fwd", dict(B=B, H=num_heads, T=T, S=S, D=D, esize=q.element_size()), ptr(q), ptr(k), ptr(v), ptr(key_valid), ptr(out), ptr(lse), B, num_heads, T, S, D,
                                   dtype_code(q), stream_ptr())
        ctx.save_for_backward(q, k, v, key_valid, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, key_valid, lse = ctx.saved_tensors
        H = ctx.num_heads
        B, T, d = q.shape
        S = k.shape[1]
        D = d // H
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        nbytes = lib().mmgl_xattn_bwd_workspace(B, H, T, S, D)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        _lib.call("mmgl_xattn_bwd", dict(B=B, H=H, T=T, S=S, D=D, esize=q.element_size()), ptr(dout), ptr(q), ptr(k), ptr(v), ptr(lse), ptr(key_valid), ptr(dq), ptr(dk), ptr(dv),
                                   ptr(ws), nbytes, B, H, T, S, D, dtype_code(q), stream_ptr())
        return dq, dk, dv, None, None


def xattn_core(q, k, v, key_valid, num_heads):
    """softmax(max(q k^T + M, finfo.min)) v per head.  q [B,T,d] is already scaled; k, v [B,S,d];
    key_valid [B,S] bool/uint8 (True = attend).  Mirrors the core of MPTAttention.forward
    (reference model/modelling_cross_attention.py:206-271)."""
    if q.dim() != 3 or k.shape != v.shape or k.dim() != 3 or q.shape[0] != k.shape[0] or q.shape[2] != k.shape[2]:
        raise ValueError(f"xattn_core: incompatible shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if q.shape[2] % num_heads:
        raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {q.shape[2]} and `num_heads`: {num_heads}).")
    if key_valid.shape != k.shape[:2]:
        raise ValueError(f"Attention mask should be of size {tuple(k.shape[:2])}, but is {tuple(key_valid.shape)}")
    if key_valid.dtype != torch.uint8:
        key_valid = key_valid.to(torch.uint8)
    return _XAttnCore.apply(q, k, v, key_valid.contiguous(), num_heads)


# ------------------------------------------------------------------------------------------ general (unfused) attention core
class _AttnGeneral(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_valid, head_mask, num_heads, causal, p_drop, seed, want_probs):
        require_cuda(q, k, v, key_valid)
        B, T, d = q.shape
        S = k.shape[1]
        D = d // num_heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = torch.empty_like(q)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=q.device)
        probs = torch.empty(B, num_heads, T, S, dtype=q.dtype, device=q.device) if want_probs else None
        hm = None if head_mask is None else head_mask.detach().to(device=q.device, dtype=torch.float32).contiguous()
        _lib.call("mmgl_attn_general_fwd", dict(flops=4.0 * B * T * S * d), ptr(q), ptr(k), ptr(v), ptr(key_valid), ptr(hm) if hm is not None else None,
                  ptr(out), ptr(probs) if probs is not None else None, ptr(lse), B, num_heads, T, S, D, int(causal), float(p_drop), int(seed),
                  dtype_code(q), stream_ptr())
        ctx.save_for_backward(q, k, v, key_valid, lse, hm)
        ctx.cfg = (num_heads, int(causal), float(p_drop), int(seed))
        if probs is not None:
            ctx.mark_non_differentiable(probs)
            return out, probs
        return out, None

    @staticmethod
    def backward(ctx, dout, _dprobs):
        q, k, v, key_valid, lse, hm = ctx.saved_tensors
        H, causal, p_drop, seed = ctx.cfg
        B, T, d = q.shape
        S = k.shape[1]
        D = d // H
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        nbytes = lib().mmgl_attn_general_bwd_workspace(B, H, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        _lib.call("mmgl_attn_general_bwd", dict(flops=10.0 * B * T * S * d), ptr(dout), ptr(q), ptr(k), ptr(v), ptr(lse), ptr(key_valid),
                  ptr(hm) if hm is not None else None, ptr(dq), ptr(dk), ptr(dv), ptr(ws), nbytes, B, H, T, S, D, causal, p_drop, seed,
                  dtype_code(q), stream_ptr())
        return dq, dk, dv, None, None, None, None, None, None, None


def attn_general(q, k, v, key_valid, num_heads, causal=False, head_mask=None, p_drop=0.0, training=False, seed=None, output_attentions=False):
    """The attention core with the options the fused kernels leave out (reference model/modelling_cross_attention.py:206-271):
    `head_mask` [H] scales the probabilities of each head (:237-244), `output_attentions` also returns them as [B,H,T,S] -- head-masked,
    before dropout (:246-254); the returned tensor is DETACHED (the reference keeps it in the graph, ":248 make sure that attn_weights
    keeps its gradient": differentiating through the returned probabilities is not supported here and autograd will say so) --, `p_drop` drops probabilities in training (:256; counter hash of (seed, index),
    regenerated in backward).  q [B,T,d] is already scaled; causal = the decoder's self-attention (S == T).  Returns (out, probs | None)."""
    if q.dim() != 3 or k.shape != v.shape or k.dim() != 3 or q.shape[0] != k.shape[0] or q.shape[2] != k.shape[2]:
        raise ValueError(f"attn_general: incompatible shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if q.shape[2] % num_heads:
        raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {q.shape[2]} and `num_heads`: {num_heads}).")
    if key_valid.shape != k.shape[:2]:
        raise ValueError(f"Attention mask should be of size {tuple(k.shape[:2])}, but is {tuple(key_valid.shape)}")
    if head_mask is not None and tuple(head_mask.shape) != (num_heads,):
        raise ValueError(f"Head mask for a single layer should be of size {(num_heads,)}, but is {tuple(head_mask.shape)}")
    if head_mask is not None and head_mask.requires_grad and torch.is_grad_enabled():
        # the reference's head mask is an ordinary multiplicand (:243): a head-importance / pruning workflow differentiates through it.
        # This kernel treats it as a constant -- refuse rather than hand back a silent zero gradient
        raise NotImplementedError("attn_general: head_mask.requires_grad is not supported (the kernel computes no gradient for the head "
                                  "mask); detach it, or differentiate a per-head scale applied outside the attention core")
    if key_valid.dtype != torch.uint8:
        key_valid = key_valid.to(torch.uint8)
    p = float(p_drop) if training else 0.0
    if p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _AttnGeneral.apply(q, k, v, key_valid.contiguous(), head_mask, num_heads, bool(causal), p, int(seed or 0), bool(output_attentions))


def attn_dropout_mask(B, H, T, S, p_drop, seed, device):
    """The keep mask attn_general uses for (p_drop, seed), as uint8 [B,H,T,S]: test / debug aid."""
    m = torch.empty(B, H, T, S, dtype=torch.uint8, device=device)
    _lib.call("mmgl_attn_dropout_mask", None, ptr(m), B, H, T, S, float(p_drop), int(seed), stream_ptr())
    return m


# ------------------------------------------------------------------------------------------ causal self-attention
class _SelfAttnCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_valid, num_heads):
        require_cuda(q, k, v, key_valid)
        B, T, d = q.shape
        D = d // num_heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = torch.empty_like(q)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=q.device)
        _lib.call("mmgl_selfattn_fwd", dict(flops=2.0 * B * T * T * d, bytes=4.0 * B * T * d * q.element_size()),
                  ptr(q), ptr(k), ptr(v), ptr(key_valid), ptr(out), ptr(lse), B, num_heads, T, D, 0, dtype_code(q), stream_ptr())
        ctx.save_for_backward(q, k, v, key_valid, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, key_valid, out, lse = ctx.saved_tensors
        H = ctx.num_heads
        B, T, d = q.shape
        D = d // H
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        nbytes = lib().mmgl_selfattn_bwd_workspace(B, H, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        _lib.call("mmgl_selfattn_bwd", dict(flops=5.0 * B * T * T * d, bytes=8.0 * B * T * d * q.element_size()),
                  ptr(dout), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(key_valid), ptr(dq), ptr(dk), ptr(dv), ptr(ws), nbytes,
                  B, H, T, D, 0, 0, dtype_code(q), stream_ptr())
        return dq, dk, dv, None, None


class _SelfAttnFusedQKV(torch.autograd.Function):
    """Same kernels, Q/K/V read in place from one fused projection output [B,T,3d] and dQ/dK/dV written into one
    [B,T,3d] buffer (row stride 3d), so the projection's dgrad is ONE GEMM and autograd adds nothing up."""

    @staticmethod
    def forward(ctx, qkv, key_valid, num_heads):
        require_cuda(qkv, key_valid)
        B, T, d3 = qkv.shape
        d = d3 // 3
        D = d // num_heads
        qkv = qkv.contiguous()
        out = torch.empty(B, T, d, dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=qkv.device)
        es = qkv.element_size()
        _lib.call("mmgl_selfattn_fwd", dict(flops=2.0 * B * T * T * d, bytes=4.0 * B * T * d * es),
                  ptr(qkv), ptr_off(qkv, d * es), ptr_off(qkv, 2 * d * es), ptr(key_valid), ptr(out), ptr(lse), B, num_heads, T, D, d3,
                  dtype_code(qkv), stream_ptr())
        ctx.save_for_backward(qkv, key_valid, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, key_valid, out, lse = ctx.saved_tensors
        H = ctx.num_heads
        B, T, d3 = qkv.shape
        d = d3 // 3
        D = d // H
        es = qkv.element_size()
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        nbytes = lib().mmgl_selfattn_bwd_workspace(B, H, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
        _lib.call("mmgl_selfattn_bwd", dict(flops=5.0 * B * T * T * d, bytes=8.0 * B * T * d * es),
                  ptr(dout), ptr(qkv), ptr_off(qkv, d * es), ptr_off(qkv, 2 * d * es), ptr(out), ptr(lse), ptr(key_valid),
                  ptr(dqkv), ptr_off(dqkv, d * es), ptr_off(dqkv, 2 * d * es), ptr(ws), nbytes, B, H, T, D, d3, d3, dtype_code(qkv), stream_ptr())
        return dqkv, None, None


def selfattn_core_fused(qkv, key_valid, num_heads):
    """selfattn_core over a fused projection output: qkv [B,T,3d] = [q*scale | k | v] along the last dim."""
    if qkv.dim() != 3 or qkv.shape[2] % (3 * num_heads):
        raise ValueError(f"selfattn_core_fused: qkv{tuple(qkv.shape)} is not [B, T, 3*H*D] for H={num_heads}")
    if key_valid.shape != qkv.shape[:2]:
        raise ValueError(f"Attention mask should be of size {tuple(qkv.shape[:2])}, but is {tuple(key_valid.shape)}")
    if key_valid.dtype != torch.uint8:
        key_valid = key_valid.to(torch.uint8)
    return _SelfAttnFusedQKV.apply(qkv, key_valid.contiguous(), num_heads)


def selfattn_core(q, k, v, key_valid, num_heads):
    """Causal self-attention softmax(mask(q k^T)) v with mask = (s <= t) & key_valid[b, s]; q is already scaled.
    The caller guarantees key_valid[:, 0] is all ones (see include/mmgl_hip.h).  (reference :203-271 self branch)"""
    if q.dim() != 3 or q.shape != k.shape or k.shape != v.shape:
        raise ValueError(f"selfattn_core: incompatible shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if q.shape[2] % num_heads:
        raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {q.shape[2]} and `num_heads`: {num_heads}).")
    if key_valid.shape != q.shape[:2]:
        raise ValueError(f"Attention mask should be of size {tuple(q.shape[:2])}, but is {tuple(key_valid.shape)}")
    if key_valid.dtype != torch.uint8:
        key_valid = key_valid.to(torch.uint8)
    return _SelfAttnCore.apply(q, k, v, key_valid.contiguous(), num_heads)


class _SelfAttnPrefix(torch.autograd.Function):
    """Causal self-attention with P always-visible prefix keys in front of the T causal ones (mmgl_selfattn_prefix_fwd/_bwd)."""

    @staticmethod
    def forward(ctx, q, k, v, key_valid, num_heads, P):
        require_cuda(q, k, v, key_valid)
        B, T, d = q.shape
        D = d // num_heads
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = torch.empty_like(q)
        lse = torch.empty(B, num_heads, T, dtype=torch.float32, device=q.device)
        _lib.call("mmgl_selfattn_prefix_fwd", dict(flops=2.0 * B * T * (T + 2 * P) * d, bytes=4.0 * B * T * d * q.element_size()),
                  ptr(q), ptr(k), ptr(v), ptr(key_valid), ptr(out), ptr(lse), B, num_heads, T, P, D, 0, 0, dtype_code(q), stream_ptr())
        ctx.save_for_backward(q, k, v, key_valid, out, lse)
        ctx.num_heads, ctx.P = num_heads, P
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, key_valid, out, lse = ctx.saved_tensors
        H, P = ctx.num_heads, ctx.P
        B, T, d = q.shape
        D = d // H
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        nbytes = lib().mmgl_selfattn_bwd_workspace(B, H, T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
        _lib.call("mmgl_selfattn_prefix_bwd", dict(flops=5.0 * B * T * (T + 2 * P) * d, bytes=8.0 * B * T * d * q.element_size()),
                  ptr(dout), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(key_valid), ptr(dq), ptr(dk), ptr(dv), ptr(ws), nbytes,
                  B, H, T, P, D, 0, 0, 0, 0, dtype_code(q), stream_ptr())
        return dq, dk, dv, None, None, None


def selfattn_core_prefix(q, k, v, key_valid, num_heads, prefix_len):
    """softmax(mask(q k^T)) v where k, v [B, P+T, d] carry P = prefix_len always-visible rows in front of the T causal ones:
    mask = (s <= t + P) & key_valid[b, s]; q [B, T, d] is already scaled.  The attention of an OPT layer under peft prefix tuning
    (reference model/modelling_self_attention.py:88-93: HF prepends the learned per-layer key/value prefix as past_key_values)."""
    if q.dim() != 3 or k.shape != v.shape or k.shape[0] != q.shape[0] or k.shape[2] != q.shape[2] or k.shape[1] != q.shape[1] + prefix_len:
        raise ValueError(f"selfattn_core_prefix: incompatible shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)} for a prefix of {prefix_len}")
    if q.shape[2] % num_heads:
        raise ValueError(f"embed_dim must be divisible by num_heads (got `embed_dim`: {q.shape[2]} and `num_heads`: {num_heads}).")
    if key_valid.shape != k.shape[:2]:
        raise ValueError(f"Attention mask should be of size {tuple(k.shape[:2])}, but is {tuple(key_valid.shape)}")
    if key_valid.dtype != torch.uint8:
        key_valid = key_valid.to(torch.uint8)
    return _SelfAttnPrefix.apply(q, k, v, key_valid.contiguous(), num_heads, int(prefix_len))


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------ LayerNorm / RMSNorm
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        require_cuda(x)
        shape = x.shape
        cols = shape[-1]
        x2 = x.contiguous().view(-1, cols)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.to(x.dtype).contiguous()
        b = None if beta is None else beta.to(x.dtype).contiguous()
        _lib.call("mmgl_layernorm_fwd", dict(bytes=2.0 * rows * cols * x.element_size()), ptr(x2), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps,
                                       dtype_code(x), stream_ptr())
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.shape = shape
        ctx.pgrad = (gamma is not None and gamma.requires_grad, beta is not None and beta.requires_grad)
        ctx.pdtype = None if gamma is None else gamma.dtype
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        rows, cols = x2.shape
        dy2 = dy.contiguous().view(rows, cols)
        dx = torch.empty_like(x2)
        want = any(ctx.pgrad)
        dgamma = torch.empty(cols, dtype=torch.float32, device=x2.device) if want else None
        dbeta = torch.empty(cols, dtype=torch.float32, device=x2.device) if want else None
        nbytes = lib().mmgl_norm_bwd_workspace(rows, cols) if want else 0
        ws = _ws(nbytes, x2.device)
        _lib.call("mmgl_layernorm_bwd", dict(bytes=3.0 * rows * cols * x2.element_size()), ptr(dy2), ptr(x2), ptr(g), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta),
                                       ptr(ws), ws.numel(), rows, cols, dtype_code(x2), stream_ptr())
        dg = dgamma.to(ctx.pdtype) if ctx.pgrad[0] else None
        db = dbeta.to(ctx.pdtype) if ctx.pgrad[1] else None
        return dx.view(ctx.shape), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    """nn.LayerNorm over the last dim (reference model/modelling_cross_attention.py:287-294, 319-320, 349-350)."""
    return _LayerNorm.apply(x, gamma, beta, float(eps))


class _LayerNormFanout(torch.autograd.Function):
    """(x, LayerNorm(x)) for a pre-LN block whose input also feeds the block's residual add (reference :316-320 `residual =
    hidden_states; hidden_states = self.self_attn_layer_norm(hidden_states)`, :347-350): the gradient arriving on the residual
    stream is added INSIDE the LayerNorm backward kernel (mmgl_add_layernorm_bwd with the sum = x), instead of autograd's
    accumulation add over [B, T, d] at every fan-out (10 per step at config 3)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        require_cuda(x)
        shape = x.shape
        cols = shape[-1]
        x2 = x.contiguous().view(-1, cols)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.to(x.dtype).contiguous()
        b = None if beta is None else beta.to(x.dtype).contiguous()
        _lib.call("mmgl_layernorm_fwd", dict(bytes=2.0 * rows * cols * x.element_size()), ptr(x2), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps,
                  dtype_code(x), stream_ptr())
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.set_materialize_grads(False)
        ctx.shape = shape
        ctx.pgrad = (gamma is not None and gamma.requires_grad, beta is not None and beta.requires_grad)
        ctx.pdtype = None if gamma is None else gamma.dtype
        return x, y.view(shape)          # x comes back as an alias of the input (autograd wraps it as a view)

    @staticmethod
    def backward(ctx, dres, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        rows, cols = x2.shape
        if dy is None:
            return dres, None, None, None
        dy2 = dy.contiguous().view(rows, cols)
        dr2 = None if dres is None else dres.contiguous().view(rows, cols)
        dx = torch.empty_like(x2)
        want = any(ctx.pgrad)
        dgamma = torch.empty(cols, dtype=torch.float32, device=x2.device) if want else None
        dbeta = torch.empty(cols, dtype=torch.float32, device=x2.device) if want else None
        nbytes = lib().mmgl_norm_bwd_workspace(rows, cols) if want else 0
        ws = _ws(nbytes, x2.device)
        nt = 3.0 + (dr2 is not None)
        _lib.call("mmgl_add_layernorm_bwd", dict(bytes=nt * rows * cols * x2.element_size()), ptr(dy2), ptr(dr2), ptr(x2), ptr(g), ptr(mean),
                  ptr(rstd), ptr(dx), None, ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), rows, cols, 0.0, 0, dtype_code(x2), stream_ptr())
        dg = dgamma.to(ctx.pdtype) if ctx.pgrad[0] else None
        db = dbeta.to(ctx.pdtype) if ctx.pgrad[1] else None
        return dx.view(ctx.shape), dg, db, None


def layer_norm_fanout(x, gamma, beta, eps=1e-5):
    """(residual, LayerNorm(x)) with residual == x: use BOTH outputs downstream (the residual add and the normed branch) so that
    the two gradients meet inside the LayerNorm backward kernel."""
    return _LayerNormFanout.apply(x, gamma, beta, float(eps))


class _AddLayerNorm(torch.autograd.Function):
    """(s, y) = (res + dropout(x), LayerNorm(s)) in one pass; backward folds the gradient arriving on s into the LayerNorm
    backward kernel, which writes the gradient of res and (when dropout is active) of x -- no separate residual kernels,
    no autograd accumulation add."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p_drop, seed):
        require_cuda(x, res)
        shape = x.shape
        cols = shape[-1]
        x2, r2 = x.contiguous().view(-1, cols), res.contiguous().view(-1, cols)
        rows = x2.shape[0]
        s, y = torch.empty_like(x2), torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.to(x.dtype).contiguous()
        b = None if beta is None else beta.to(x.dtype).contiguous()
        _lib.call("mmgl_add_layernorm_fwd", dict(bytes=4.0 * rows * cols * x.element_size()), ptr(x2), ptr(r2), ptr(g), ptr(b), ptr(s), ptr(y),
                  ptr(mean), ptr(rstd), rows, cols, eps, p_drop, seed, dtype_code(x), stream_ptr())
        ctx.save_for_backward(s, g, mean, rstd)
        ctx.set_materialize_grads(False)                      # an unused output arrives as None, not as a zero tensor to stream
        ctx.shape, ctx.p, ctx.seed = shape, p_drop, seed
        ctx.pgrad = (gamma is not None and gamma.requires_grad, beta is not None and beta.requires_grad)
        ctx.pdtype = None if gamma is None else gamma.dtype
        return s.view(shape), y.view(shape)

    @staticmethod
    def backward(ctx, ds, dy):
        s, g, mean, rstd = ctx.saved_tensors
        rows, cols = s.shape
        p = ctx.p
        if dy is None:                                        # only the residual stream was used downstream
            if ds is None:
                return None, None, None, None, None, None, None
            if p == 0.0:
                return ds, ds, None, None, None, None, None
            dy = torch.zeros_like(ds)
        dy2 = dy.contiguous().view(rows, cols)
        ds2 = None if ds is None else ds.contiguous().view(rows, cols)
        dres = torch.empty_like(s)
        dx = torch.empty_like(s) if p > 0.0 else None
        want = any(ctx.pgrad)
        dgamma = torch.empty(cols, dtype=torch.float32, device=s.device) if want else None
        dbeta = torch.empty(cols, dtype=torch.float32, device=s.device) if want else None
        nbytes = lib().mmgl_norm_bwd_workspace(rows, cols) if want else 0
        ws = _ws(nbytes, s.device)
        nt = 3.0 + (ds2 is not None) + (dx is not None)
        _lib.call("mmgl_add_layernorm_bwd", dict(bytes=nt * rows * cols * s.element_size()), ptr(dy2), ptr(ds2), ptr(s), ptr(g), ptr(mean),
                  ptr(rstd), ptr(dres), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel(), rows, cols, p, ctx.seed, dtype_code(s),
                  stream_ptr())
        dres = dres.view(ctx.shape)
        dx = dres if dx is None else dx.view(ctx.shape)
        dg = dgamma.to(ctx.pdtype) if ctx.pgrad[0] else None
        db = dbeta.to(ctx.pdtype) if ctx.pgrad[1] else None
        return dx, dres, dg, db, None, None, None


def add_layer_norm_pair(x, res, gamma, beta, eps=1e-5, p_drop=0.0, training=False, seed=None):
    """Differentiable (s, LayerNorm(s)) with s = res + dropout(x): the `hidden = residual + dropout(h)` / next-LayerNorm
    pair of a decoder layer (reference model/modelling_cross_attention.py:332-350) as one forward and one backward
    kernel.  Dropout uses the same counter hash of (seed, element index) as gated_residual."""
    if x.shape != res.shape:
        raise ValueError(f"add_layer_norm_pair: shapes differ {tuple(x.shape)} vs {tuple(res.shape)}")
    p = float(p_drop) if training else 0.0
    if p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _AddLayerNorm.apply(x, res, gamma, beta, float(eps), p, int(seed or 0))


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, eps):
        require_cuda(x)
        shape = x.shape
        cols = shape[-1]
        x2 = x.contiguous().view(-1, cols)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.to(x.dtype).contiguous()
        _lib.call("mmgl_rmsnorm_fwd", None, ptr(x2), ptr(g), ptr(y), ptr(rstd), rows, cols, eps, dtype_code(x), stream_ptr())
        ctx.save_for_backward(x2, g, rstd)
        ctx.shape = shape
        ctx.pgrad = gamma is not None and gamma.requires_grad
        ctx.pdtype = None if gamma is None else gamma.dtype
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, rstd = ctx.saved_tensors
        rows, cols = x2.shape
        dy2 = dy.contiguous().view(rows, cols)
        dx = torch.empty_like(x2)
        dgamma = torch.empty(cols, dtype=torch.float32, device=x2.device) if ctx.pgrad else None
        nbytes = lib().mmgl_norm_bwd_workspace(rows, cols) if ctx.pgrad else 0
        ws = _ws(nbytes, x2.device)
        _lib.call("mmgl_rmsnorm_bwd", None, ptr(dy2), ptr(x2), ptr(g), ptr(rstd), ptr(dx), ptr(dgamma), ptr(ws), ws.numel(), rows,
                                     cols, dtype_code(x2), stream_ptr())
        return dx.view(ctx.shape), (dgamma.to(ctx.pdtype) if ctx.pgrad else None), None


def rms_norm(x, gamma, eps=1e-6):
    return _RMSNorm.apply(x, gamma, float(eps))


class _AddRMSNorm(torch.autograd.Function):
    """(s, y) = (res + x, RMSNorm(s)) in one pass (a Llama layer's residual add and the norm that follows it); backward folds the
    gradient arriving on s into the norm's backward kernel: no residual kernel, no autograd accumulation add."""

    @staticmethod
    def forward(ctx, x, res, gamma, eps):
        require_cuda(x, res)
        shape = x.shape
        cols = shape[-1]
        x2, r2 = x.contiguous().view(-1, cols), res.contiguous().view(-1, cols)
        rows = x2.shape[0]
        s, y = torch.empty_like(x2), torch.empty_like(x2)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        g = None if gamma is None else gamma.to(x.dtype).contiguous()
        _lib.call("mmgl_add_rmsnorm_fwd", dict(bytes=4.0 * rows * cols * x.element_size()), ptr(x2), ptr(r2), ptr(g), ptr(s), ptr(y), ptr(rstd),
                  rows, cols, eps, dtype_code(x), stream_ptr())
        ctx.save_for_backward(s, g, rstd)
        ctx.set_materialize_grads(False)
        ctx.shape = shape
        ctx.pgrad = gamma is not None and gamma.requires_grad
        ctx.pdtype = None if gamma is None else gamma.dtype
        return s.view(shape), y.view(shape)

    @staticmethod
    def backward(ctx, ds, dy):
        s, g, rstd = ctx.saved_tensors
        rows, cols = s.shape
        if dy is None:                                        # only the residual stream was used downstream
            return ds, ds, None, None
        dy2 = dy.contiguous().view(rows, cols)
        ds2 = None if ds is None else ds.contiguous().view(rows, cols)
        dres = torch.empty_like(s)
        dgamma = torch.empty(cols, dtype=torch.float32, device=s.device) if ctx.pgrad else None
        ws = _ws(lib().mmgl_norm_bwd_workspace(rows, cols) if ctx.pgrad else 0, s.device)
        _lib.call("mmgl_add_rmsnorm_bwd", dict(bytes=(3.0 + (ds2 is not None)) * rows * cols * s.element_size()), ptr(dy2), ptr(ds2), ptr(s), ptr(g),
                  ptr(rstd), ptr(dres), ptr(dgamma), ptr(ws), ws.numel(), rows, cols, dtype_code(s), stream_ptr())
        dres = dres.view(ctx.shape)
        return dres, dres, (dgamma.to(ctx.pdtype) if dgamma is not None else None), None


def add_rms_norm_pair(x, res, gamma, eps=1e-6):
    """Differentiable (s, RMSNorm(s)) with s = res + x: one forward and one backward kernel for the pair."""
    if x.shape != res.shape:
        raise ValueError(f"add_rms_norm_pair: shapes differ {tuple(x.shape)} vs {tuple(res.shape)}")
    return _AddRMSNorm.apply(x, res, gamma, float(eps))


# ------------------------------------------------------------------------------------------ gated residual
class _GatedResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, residual, x, gate, p_drop, seed):
        require_cuda(residual, x)
        residual, x = residual.contiguous(), x.contiguous()
        y = torch.empty_like(x)
        g32 = None if gate is None else gate.detach().to(torch.float32).reshape(1).contiguous()
        _lib.call("mmgl_gated_residual_fwd", dict(bytes=3.0 * x.numel() * x.element_size()), ptr(residual), ptr(x), ptr(g32), ptr(y), x.numel(), p_drop, seed,
                                            dtype_code(x), stream_ptr())
        ctx.save_for_backward(x, g32)
        ctx.p, ctx.seed = p_drop, seed
        ctx.gate_meta = None if gate is None else (gate.dtype, gate.shape, gate.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgate = torch.zeros(1, dtype=torch.float32, device=x.device) if g32 is not None else None
        ws = _ws(lib().mmgl_gated_residual_bwd_workspace(x.numel()), x.device)
        _lib.call("mmgl_gated_residual_bwd", dict(bytes=3.0 * x.numel() * x.element_size()), ptr(dy), ptr(x), ptr(g32), ptr(dx), ptr(dgate), ptr(ws), ws.numel(), x.numel(),
                                            ctx.p, ctx.seed, dtype_code(x), stream_ptr())
        dg = None
        if ctx.gate_meta is not None and ctx.gate_meta[2]:
            dg = dgate.to(ctx.gate_meta[0]).reshape(ctx.gate_meta[1])
        return dy, dx, dg, None, None


def gated_residual(residual, x, gate=None, p_drop=0.0, training=False, seed=None):
    """residual + tanh(gate) * dropout(x)  (reference :332-335, :356-359; gate=None is the ungated :337/:361 form).
    The dropout mask is a counter hash of (seed, index) regenerated in backward."""
    p = float(p_drop) if training else 0.0
    if p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _GatedResidual.apply(residual, x, gate, p, int(seed or 0))


# ------------------------------------------------------------------------------------------ linear (+bias, scale, ReLU)
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, out_scale, mask_dx=False, premasked=False):
        require_cuda(x, weight)
        shape = x.shape
        K = shape[-1]
        N = weight.shape[0]
        x2 = x.contiguous().view(-1, K)
        M = x2.shape[0]
        w = weight.to(x.dtype).contiguous()
        b = None if bias is None else bias.to(x.dtype).contiguous()
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        _lib.call("mmgl_linear_fwd", dict(flops=2.0 * M * N * K, bytes=float(M * K + N * K + M * N) * x.element_size()), ptr(x2), ptr(w), ptr(b), ptr(y), M, N, K, act, out_scale, dtype_code(x), stream_ptr())
        # premasked: the consumer of y folds this layer's ReLU backward into its own dgrad (mask_dx there), so the incoming
        # gradient is already dy * (y > 0): differentiate as a plain linear and do not keep y for the mask
        ctx.save_for_backward(x2, w, y if (act and not premasked) else None)
        ctx.meta = (shape, 0 if premasked else act, out_scale, weight.dtype, None if bias is None else bias.dtype)
        ctx.mask_dx = bool(mask_dx)
        ctx.need = (x.requires_grad, weight.requires_grad, bias is not None and bias.requires_grad)
        return y.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        shape, act, out_scale, wdt, bdt = ctx.meta
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.contiguous().view(M, N)
        code = dtype_code(x2)
        dx = torch.empty_like(x2) if ctx.need[0] else None
        want_w = ctx.need[1] or ctx.need[2]
        dw = torch.empty_like(w) if want_w else None
        db = torch.empty(N, dtype=x2.dtype, device=x2.device) if ctx.need[2] else None
        dx_done = None
        if (dx is not None and N % 128 and act == _lib.ACT_NONE and not ctx.mask_dx and x2.dtype == torch.bfloat16 and K % 8 == 0
                and N % 8 == 0):
            # a contraction length that is no multiple of 128 (a trainable lm_head: N = vocab = 50272) has no large-tile dgrad
            # in mmgl_linear_bwd; W^T zero-padded to the next multiple (built per call: the weight is being trained) puts it on
            # the persistent kernel, dy read with its own row stride (frozen_dgrad's trick)
            npad = N + (-N) % 128
            if lib().mmgl_gemm_nt_fast(M, K, npad, N, npad, K, code):
                wt = torch.zeros(K, npad, dtype=w.dtype, device=w.device)
                wt[:, :N] = w.t()
                gemm_nt(dy2, wt, out_scale=out_scale, K=npad, out=dx)
                dx_done, dx = dx, None
        ws = _ws(lib().mmgl_linear_bwd_workspace(M, N, K, act, code), x2.device)
        _lib.call("mmgl_linear_bwd", dict(flops=2.0 * M * N * K * (int(dx is not None) + int(dw is not None)),
                                         bytes=float(M * K + N * K + M * N) * x2.element_size()),
                  ptr(dy2), ptr(y), ptr(x2), ptr(w), ptr(dx), ptr(dw), ptr(db), ptr(ws), ws.numel(), M, N, K, act, out_scale, 0,
                  int(ctx.mask_dx), code, stream_ptr())
        if dx_done is not None:
            dx = dx_done
        if dx is not None:
            dx = dx.view(shape)
        dw = dw.to(wdt) if (dw is not None and ctx.need[1]) else None
        db = db.to(bdt) if db is not None else None
        return dx, dw, db, None, None, None, None


def linear(x, weight, bias=None, act="none", out_scale=1.0, mask_dx=False, bwd_premasked=False):
    """act((x @ weight.T + bias) * out_scale); weight is nn.Linear layout [out, in].
    (reference q/k/v/out_proj :194-199, :273; fc1+ReLU / fc2 :352-355)
    mask_dx / bwd_premasked: a ReLU linear followed directly by another linear can hand its ReLU backward to the consumer:
    `h = linear(x, W1, b1, act="relu", bwd_premasked=True); y = linear(h, W2, b2, mask_dx=True)` -- the second call zeroes
    its dx where h <= 0 inside the dgrad GEMM, the first one then skips the separate mask pass.  Use the two flags together."""
    if x.shape[-1] != weight.shape[1]:
        raise ValueError(f"linear: x has {x.shape[-1]} features, weight expects {weight.shape[1]}")
    code = {"none": _lib.ACT_NONE, "relu": _lib.ACT_RELU}.get(act)
    if code is None:
        raise ValueError(f"linear: activation {act!r} is not fused (supported: none, relu)")
    # The MFMA kernels move 16-byte chunks along K and 4 outputs per lane along N: odd feature counts (e.g. the
    # Laplacian-PE input width k = N_neighbors - 4) are zero-padded here; autograd slices the gradients back.
    K, N = weight.shape[1], weight.shape[0]
    kq = 8 if x.dtype == torch.bfloat16 else 4
    pk, pn = (-K) % kq, (-N) % 8
    if pk or pn:
        x = torch.nn.functional.pad(x, (0, pk)) if pk else x
        weight = torch.nn.functional.pad(weight, (0, pk, 0, pn))
        if bias is not None and pn:
            bias = torch.nn.functional.pad(bias, (0, pn))
        y = _Linear.apply(x, weight, bias, code, float(out_scale), bool(mask_dx), bool(bwd_premasked))
        return y[..., :N] if pn else y
    return _Linear.apply(x, weight, bias, code, float(out_scale), bool(mask_dx), bool(bwd_premasked))


class _LoraLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, A, Bm, scale):
        require_cuda(x, weight, A, Bm)
        shape = x.shape
        K = shape[-1]
        N, r = weight.shape[0], A.shape[0]
        x2 = x.contiguous().view(-1, K)
        M = x2.shape[0]
        w, a, bm = (t.to(x.dtype).contiguous() for t in (weight, A, Bm))
        b = None if bias is None else bias.to(x.dtype).contiguous()
        y = torch.empty(M, N, dtype=x.dtype, device=x.device)
        xa = torch.empty(M, r, dtype=x.dtype, device=x.device)
        _lib.call("mmgl_lora_linear_fwd", dict(flops=2.0 * M * N * K + 2.0 * M * r * (K + N)), ptr(x2), ptr(w), ptr(b), ptr(a), ptr(bm), ptr(y), ptr(xa), M, N, K, r, scale,
                                         dtype_code(x), stream_ptr())
        ctx.save_for_backward(x2, xa, w, a, bm)
        ctx.meta = (shape, scale, A.dtype, Bm.dtype)
        return y.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, xa, w, a, bm = ctx.saved_tensors
        shape, scale, adt, bdt = ctx.meta
        M, K = x2.shape
        N, r = w.shape[0], a.shape[0]
        dy2 = dy.contiguous().view(M, N)
        code = dtype_code(x2)
        dx, dA, dB = torch.empty_like(x2), torch.empty_like(a), torch.empty_like(bm)
        dyb = torch.empty(M, r, dtype=x2.dtype, device=x2.device)
        ws = _ws(lib().mmgl_lora_linear_bwd_workspace(M, N, K, r, code), x2.device)
        _lib.call("mmgl_lora_linear_bwd", dict(flops=2.0 * M * N * K + 6.0 * M * r * (K + N)), ptr(dy2), ptr(x2), ptr(xa), ptr(w), ptr(a), ptr(bm), ptr(dx), ptr(dA), ptr(dB), ptr(dyb),
                                         ptr(ws), ws.numel(), M, N, K, r, scale, 0, code, stream_ptr())
        return dx.view(shape), None, None, dA.to(adt), dB.to(bdt), None


def _wgrad_only(dy2, x2, w_like, scale):
    """dW [N, K] = scale * dy2^T x2 through mmgl_linear_bwd (no dx, no bias gradient); w_like: any [N, K] tensor of the dtype."""
    M, K = x2.shape
    N = dy2.shape[1]
    code = dtype_code(x2)
    dw = torch.empty(N, K, dtype=x2.dtype, device=x2.device)
    ws = _ws(lib().mmgl_linear_bwd_workspace(M, N, K, 0, code), x2.device)
    _lib.call("mmgl_linear_bwd", dict(flops=2.0 * M * N * K, bytes=float(M * K + N * K + M * N) * x2.element_size()),
              ptr(dy2), None, ptr(x2), ptr(w_like), None, ptr(dw), None, ptr(ws), ws.numel(), M, N, K, 0, float(scale), 0, 0, code, stream_ptr())
    return dw




class _LoraLinearBig(torch.autograd.Function):
    """lora_linear for large bf16 shapes as ONE autograd node: the same seven GEMMs the composed form ran (rank zero-padded to
    256), but dx = dy W + (dy B) A leaves the second GEMM's residual epilogue instead of an ATen add, the pads / slices of the
    adapter gradients happen once, and nothing but x, x A^T and the padded factors is kept (config 4: 201 adds and 123 copies
    per step came from autograd's bookkeeping around the composed form)."""

    @staticmethod
    def forward(ctx, x, weight, bias, lora_A, lora_B, scale, out_scale=1.0):
        require_cuda(x, weight)
        K, N, r = x.shape[-1], weight.shape[0], lora_A.shape[0]
        rp = (-r) % 256
        A_pad = F.pad(lora_A.detach().to(x.dtype), (0, 0, 0, rp)).contiguous()          # [256, K]
        B_pad = F.pad(lora_B.detach().to(x.dtype), (0, rp)).contiguous()                # [N, 256]
        x2 = x.reshape(-1, K).contiguous()
        xa = gemm_nt(x2, A_pad)                                                         # [M, 256]
        delta = gemm_nt(xa, B_pad, out_scale=scale * out_scale)                         # [M, N]  (out_scale: attention's D^-1/2 on q_proj)
        w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
        b = None if bias is None else (bias if bias.dtype == x.dtype else bias.to(x.dtype))
        out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
        gemm_nt(x2, w.contiguous(), b, residual=delta, out_scale=out_scale, out=out.view(-1, N))
        ctx.save_for_backward(x2, xa, weight, A_pad, B_pad)
        ctx.meta = (x.shape, float(scale), r, lora_A.dtype, lora_B.dtype, float(out_scale))
        return out

    @staticmethod
    def backward(ctx, dy):
        x2, xa, weight, A_pad, B_pad = ctx.saved_tensors
        xshape, scale, r, adt, bdt, os_ = ctx.meta
        N = weight.shape[0]
        g = dy.reshape(-1, N).contiguous()
        dxa = gemm_nt(g, B_pad.t().contiguous(), out_scale=scale * os_)                 # (dy B) * scale  [M, 256]
        dB = _wgrad_only(g, xa, B_pad, scale * os_) if ctx.needs_input_grad[4] else None      # [N, 256]
        dA = _wgrad_only(dxa, x2, A_pad, 1.0) if ctx.needs_input_grad[3] else None      # [256, K]
        dx = None
        if ctx.needs_input_grad[0]:
            low = gemm_nt(dxa, A_pad.t().contiguous())                                  # (dy B) A  [M, K]
            dx = frozen_dgrad(g, weight, out_scale=os_, residual=low).view(xshape)      # dy W * out_scale + ... in the epilogue
        return (dx, None, None, None if dA is None else dA[:r].to(adt), None if dB is None else dB[:, :r].contiguous().to(bdt), None, None)


class _LoraQKV(torch.autograd.Function):
    """q | k | v of a self-attention layer whose q_proj and v_proj carry LoRA adapters over frozen weights and whose k_proj is
    frozen (peft's default targets for OPT, reference model/modelling_self_attention.py:80-87), as ONE autograd node over the
    layer's fused [3d, K] weight: forward = x [A_q ; A_v]^T (one skinny GEMM, both adapters in one rank-padded factor), the fused
    base GEMM, and the two low-rank updates accumulated IN PLACE into the q and v column blocks (residual = output); backward, from
    the attention kernels' fused dq | dk | dv buffer = (dqkv B_cat) in one GEMM, both adapters' dB / dA in one weight-gradient
    GEMM each, and dx = dqkv W_qkv + (dqkv B_cat) A_cat as ONE dgrad GEMM with the low-rank term in its epilogue.  Against three
    separate projections: 5 GEMMs instead of 11 in backward, 4 instead of 7 forward, and none of autograd's two gradient adds per
    layer (x feeds three nodes there)."""

    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, Aq, Bq, Av, Bv, scale, q_scale):
        require_cuda(x, w_qkv)
        K, d, r = x.shape[-1], w_qkv.shape[0] // 3, Aq.shape[0]
        dt, dev = x.dtype, x.device
        x2 = x.reshape(-1, K).contiguous()
        M = x2.shape[0]
        A_cat = torch.zeros(256, K, dtype=dt, device=dev)                               # rows 0..r-1: A_q, r..2r-1: A_v
        A_cat[:r], A_cat[r:2 * r] = Aq.detach().to(dt), Av.detach().to(dt)
        B_cat = torch.zeros(3 * d, 256, dtype=dt, device=dev)                           # [q rows | k rows = 0 | v rows] x [q ranks | v ranks]
        B_cat[:d, :r] = Bq.detach().float().mul(scale * q_scale).to(dt)
        B_cat[2 * d:, r:2 * r] = Bv.detach().float().mul(scale).to(dt)
        xa = gemm_nt(x2, A_cat)                                                         # [M, 256]
        qkv = torch.empty(*x.shape[:-1], 3 * d, dtype=dt, device=dev)
        q2 = qkv.view(M, 3 * d)
        gemm_nt(x2, w_qkv, b_qkv, out=q2)
        gemm_nt(xa, B_cat[:d], residual=q2[:, :d], out=q2[:, :d])                       # q += s qs (x A_q^T) B_q^T   (in place)
        gemm_nt(xa, B_cat[2 * d:], residual=q2[:, 2 * d:], out=q2[:, 2 * d:])           # v += s (x A_v^T) B_v^T
        ctx.save_for_backward(x2, xa, w_qkv, A_cat, B_cat)
        ctx.meta = (x.shape, r, float(scale), float(q_scale), Aq.dtype, Bq.dtype)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        x2, xa, w_qkv, A_cat, B_cat = ctx.saved_tensors
        xshape, r, scale, q_scale, adt, bdt = ctx.meta
        d = w_qkv.shape[0] // 3
        g = dqkv.reshape(-1, 3 * d).contiguous()
        dxa = gemm_nt(g, B_cat.t().contiguous())                                        # [M, 256]: (dq B_q) s qs | (dv B_v) s
        need = ctx.needs_input_grad
        dAq = dAv = dBq = dBv = None
        if need[3] or need[5]:
            dA = _wgrad_only(dxa, x2, A_cat, 1.0)                                       # [256, K]
            dAq, dAv = dA[:r].to(adt), dA[r:2 * r].to(adt)
        if need[4] or need[6]:
            dB = _wgrad_only(g, xa, B_cat, scale)                                       # [3d, 256] = s dqkv^T (x A_cat^T)
            dBq = (dB[:d, :r].float() * q_scale).to(bdt)
            dBv = dB[2 * d:, r:2 * r].contiguous().to(bdt)
        dx = None
        if need[0]:
            low = gemm_nt(dxa, A_cat.t().contiguous())                                  # [M, K]
            dx = frozen_dgrad(g, w_qkv, residual=low).view(xshape)
        return dx, None, None, dAq, dBq, dAv, dBv, None, None


def lora_qkv_supported(x, w_qkv, r):
    """The fused LoRA q | k | v node takes bf16 CUDA activations, both adapters in one 256-wide factor, and a shape whose base GEMM
    runs on the persistent kernel (anything else: the three projections one by one)."""
    M, K = x.numel() // x.shape[-1], x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and w_qkv.dtype == torch.bfloat16 and 2 * r <= 256 and K % 8 == 0 and w_qkv.shape[0] % 24 == 0
            and lib().mmgl_gemm_nt_fast(M, w_qkv.shape[0], K, K, K, w_qkv.shape[0], _lib.BF16) == 1
            and lib().mmgl_gemm_nt_fast(M, w_qkv.shape[0] // 3, 256, 256, 256, w_qkv.shape[0], _lib.BF16) == 1)


def lora_qkv(x, w_qkv, b_qkv, lora_Aq, lora_Bq, lora_Av, lora_Bv, scale, q_scale):
    """[..., 3d] = (q * q_scale | k | v) with q = x W_q^T + b_q + scale (x A_q^T) B_q^T, v likewise, k = x W_k^T + b_k; w_qkv / b_qkv
    are the layer's frozen fused weight and bias with q_scale already folded into the q rows.  Check lora_qkv_supported first."""
    return _LoraQKV.apply(x, w_qkv, b_qkv, lora_Aq, lora_Bq, lora_Av, lora_Bv, float(scale), float(q_scale))


def lora_linear(x, weight, bias, lora_A, lora_B, scale, out_scale=1.0):
    """x W^T + b + scale * (x A^T) B^T with a frozen base weight (peft LoRA semantics, lora_dropout = 0).
    Large bf16 shapes: the base product runs on the persistent ping-pong GEMM with the low-rank update
    delta = (x A^T)(scale B)^T -- two skinny, HBM-bound GEMMs -- added in its epilogue; backward = the frozen dgrad GEMM plus
    the four skinny products autograd derives from the same pieces.  Everything else: one fused kernel (mmgl_lora_linear_*)
    that carries the rank-r term as a second operand pair in the same accumulators."""
    M = x.numel() // x.shape[-1]
    N, K = weight.shape
    if (not weight.requires_grad and (bias is None or not bias.requires_grad) and x.is_cuda
            and lib().mmgl_gemm_nt_fast(M, N, K, K, K, N, dtype_code(x)) == 1):
        # the rank is zero-padded to 256 (autograd slices the gradients back): every product of the low-rank path -- x A^T,
        # (x A^T) B^T, and in backward dy B, (dy B) A, (dy B)^T x, dy^T (x A^T) -- is then a 256-wide GEMM the large-tile kernels
        # (and their K splits) carry, instead of a 16-wide one on a handful of workgroups
        if x.dtype == torch.bfloat16 and K % 8 == 0 and N % 8 == 0:
            return _LoraLinearBig.apply(x, weight, bias, lora_A, lora_B, float(scale), float(out_scale))
        if out_scale != 1.0:
            return lora_linear(x, weight, bias, lora_A, lora_B, scale) * out_scale
        rp = (-lora_A.shape[0]) % 256
        A_pad = F.pad(lora_A, (0, 0, 0, rp)) if rp else lora_A
        B_pad = F.pad(lora_B, (0, rp)) if rp else lora_B
        xa = linear(x, A_pad)                                      # [.., 256]
        delta = linear(xa, B_pad, out_scale=scale)                 # [.., N]
        return frozen_linear(x, weight, bias, residual=delta)
    y = _LoraLinear.apply(x, weight, bias, lora_A, lora_B, float(scale))
    return y if out_scale == 1.0 else y * out_scale


# ------------------------------------------------------------------------------------------ neighbor interleave
class _Interleave(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text_emb, vis_emb, text_loc, img_loc, text_pos, img_pos):
        require_cuda(text_emb)
        B, Nt, n_tok, d = text_emb.shape
        Ni = 0 if vis_emb is None else vis_emb.shape[1]
        text_emb = text_emb.contiguous()
        vis = None if vis_emb is None else vis_emb.contiguous()
        out = torch.empty(B, (Nt + Ni) * n_tok, d, dtype=text_emb.dtype, device=text_emb.device)
        valid = torch.empty(B, (Nt + Ni) * n_tok, dtype=torch.uint8, device=text_emb.device)
        tl, tp = text_loc.contiguous(), text_pos.contiguous()
        il = None if img_loc is None else img_loc.contiguous()
        ip = None if img_pos is None else img_pos.contiguous()
        _lib.call("mmgl_neighbor_interleave_fwd", None, ptr(text_emb), ptr(vis), ptr(tl), ptr(il), ptr(tp), ptr(ip), ptr(out),
                                                 ptr(valid), B, Nt, Ni, n_tok, d, dtype_code(text_emb), stream_ptr())
        ctx.save_for_backward(tl, il)
        ctx.dims = (B, Nt, Ni, n_tok, d)
        ctx.mark_non_differentiable(valid)
        return out, valid

    @staticmethod
    def backward(ctx, dout, _dvalid):
        tl, il = ctx.saved_tensors
        B, Nt, Ni, n_tok, d = ctx.dims
        dout = dout.contiguous()
        dtext = torch.empty(B, Nt, n_tok, d, dtype=dout.dtype, device=dout.device)
        dvis = torch.empty(B, Ni, n_tok, d, dtype=dout.dtype, device=dout.device) if Ni else None
        _lib.call("mmgl_neighbor_interleave_bwd", None, ptr(dout), ptr(tl), ptr(il), ptr(dtext), ptr(dvis), B, Nt, Ni, n_tok, d,
                                                 dtype_code(dout), stream_ptr())
        return dtext, dvis, None, None, None, None


def neighbor_interleave(text_emb, vis_emb, text_loc, img_loc, text_pos, img_pos):
    """Scatter [B,Nt,n,d] text and [B,Ni,n,d] image neighbor tokens into slot order; returns
    (neighbor_embeds [B,(Nt+Ni)*n,d], key_valid uint8 [B,(Nt+Ni)*n])   (reference :1080-1104)."""
    if vis_emb is not None and vis_emb.shape[2:] != text_emb.shape[2:]:
        raise ValueError("n_text_tokens must equal n_visual_tokens for the interleaved layout (reference :1083-1098)")
    return _Interleave.apply(text_emb, vis_emb, text_loc, img_loc, text_pos, img_pos)


# ------------------------------------------------------------------------------------------ cross entropy
class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        require_cuda(logits, labels)
        rows, V = logits.shape
        logits = logits.contiguous()
        labels = labels.contiguous()
        dev = logits.device
        row_lse = torch.empty(rows, dtype=torch.float32, device=dev)
        row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
        sums = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.call("mmgl_cross_entropy_fwd", dict(bytes=1.0 * rows * V * logits.element_size()), ptr(logits), ptr(labels), ptr(row_lse), ptr(row_loss), ptr(sums[0:1]), ptr(sums[1:2]),
                                           rows, V, ignore_index, dtype_code(logits), stream_ptr())
        ctx.save_for_backward(logits, labels, row_lse, sums)
        ctx.ignore = ignore_index
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, dloss):
        logits, labels, row_lse, sums = ctx.saved_tensors
        rows, V = logits.shape
        dl = dloss.detach().to(torch.float32).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        _lib.call("mmgl_cross_entropy_bwd", dict(bytes=2.0 * rows * V * logits.element_size()), ptr(logits), ptr(labels), ptr(row_lse), ptr(sums[1:2]), ptr(dl), ptr(dlogits), rows, V,
                                           ctx.ignore, dtype_code(logits), stream_ptr())
        return dlogits, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """Mean token cross-entropy of [rows, V] logits (fp32 scalar), nn.CrossEntropyLoss semantics (reference :831-836)."""
    return _CrossEntropy.apply(logits, labels, int(ignore_index))


class _LMHeadCrossEntropy(torch.autograd.Function):
    """Token cross-entropy of lm_head(hidden) WITHOUT the [rows, V] logits / dlogits tensors (4.1 GB each at B=64): rows are
    processed in chunks through one scratch buffer -- logits chunk = GEMM, CE statistics, dlogits chunk in place, d hidden
    chunk = GEMM against the cached W^T -- all in the FORWARD pass for a unit upstream gradient; backward scales d hidden by
    the incoming scalar.  Same FLOPs as the unfused path (no recomputation).  The head is frozen (tied to the frozen
    embedding in every peft mode of the reference, :731-737)."""

    @staticmethod
    def forward(ctx, hidden, weight, labels, ignore_index, chunk_rows):
        require_cuda(hidden, weight, labels)
        V, d = weight.shape
        h2 = hidden.reshape(-1, d)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        M = h2.shape[0]
        lab = labels.reshape(-1).contiguous()
        dev = h2.device
        w = weight if weight.dtype == h2.dtype else weight.to(h2.dtype)
        count = (lab != ignore_index).sum().to(torch.float32).reshape(1)
        one = torch.ones(1, dtype=torch.float32, device=dev)
        loss_sum = torch.zeros(1, dtype=torch.float32, device=dev)
        dh = torch.empty_like(h2)
        chunk = min(M, int(chunk_rows))
        scratch = torch.empty(chunk, V, dtype=h2.dtype, device=dev)
        row_lse = torch.empty(chunk, dtype=torch.float32, device=dev)
        row_loss = torch.empty(chunk, dtype=torch.float32, device=dev)
        part = torch.empty(2, dtype=torch.float32, device=dev)
        code = dtype_code(h2)
        es = h2.element_size()
        for r0 in range(0, M, chunk):
            r1 = min(M, r0 + chunk)
            n = r1 - r0
            lg = scratch[:n]
            gemm_nt(h2[r0:r1], w.contiguous(), out=lg)
            _lib.call("mmgl_cross_entropy_fwd", dict(bytes=1.0 * n * V * es), ptr(lg), ptr(lab[r0:r1]), ptr(row_lse), ptr(row_loss), ptr(part[0:1]),
                      ptr(part[1:2]), n, V, ignore_index, code, stream_ptr())
            loss_sum += part[0:1]
            _lib.call("mmgl_cross_entropy_bwd", dict(bytes=2.0 * n * V * es), ptr(lg), ptr(lab[r0:r1]), ptr(row_lse), ptr(count), ptr(one), ptr(lg), n, V,
                      ignore_index, code, stream_ptr())
            frozen_dgrad(lg, weight, out=dh[r0:r1])
        ctx.save_for_backward(dh)
        ctx.shape = hidden.shape
        return (loss_sum / count.clamp_min(1.0)).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dh,) = ctx.saved_tensors
        # scale in fp32 by the device scalar (mmgl_scale: one pass, no host sync): 1 / grad_accumulation_steps rounded to bf16
        # first would bias every hidden gradient by up to 0.4 %
        s = dloss.detach().to(torch.float32).reshape(1).contiguous()
        out = torch.empty_like(dh)
        _lib.call("mmgl_scale", dict(bytes=2.0 * dh.numel() * dh.element_size()), ptr(dh), ptr(s), ptr(out), dh.numel(), dtype_code(dh),
                  stream_ptr())
        return out.view(ctx.shape), None, None, None, None


def lm_head_cross_entropy(hidden, weight, labels, ignore_index=-100, chunk_rows=8192):
    """Mean token cross-entropy of `hidden @ weight.T` against `labels` (nn.CrossEntropyLoss semantics) with a FROZEN head,
    never materialising the logits (reference :826-836 builds [B, T, V] logits, the shifted copy and their gradients).
    hidden [..., d], weight [V, d], labels [...] int64 (already shifted).  fp32 scalar."""
    if weight.requires_grad:
        raise ValueError("lm_head_cross_entropy: the head must be frozen (use lm_head + cross_entropy for a trainable head)")
    if hidden.shape[:-1] != labels.shape or hidden.shape[-1] != weight.shape[1]:
        raise ValueError(f"lm_head_cross_entropy: shapes hidden{tuple(hidden.shape)} weight{tuple(weight.shape)} labels{tuple(labels.shape)}")
    return _LMHeadCrossEntropy.apply(hidden, weight, labels, int(ignore_index), int(chunk_rows))


def position_ids(attention_mask):
    """cumsum(mask) * mask - 1 + 2  (reference MPTLearnedPositionalEmbedding :135-145)."""
    require_cuda(attention_mask)
    m = attention_mask.to(torch.int64).contiguous()
    out = torch.empty_like(m)
    _lib.call("mmgl_position_ids", None, ptr(m), ptr(out), m.shape[0], m.shape[1], stream_ptr())
    return out


def adamw_step_(param, master, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    """In-place fused AdamW over flat buffers (torch.optim.AdamW formula)."""
    require_cuda(param, grad)
    _lib.call("mmgl_adamw_step", None, ptr(param), ptr(master), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, beta1, beta2,
                                eps, weight_decay, step, grad_scale, dtype_code(param), stream_ptr())


# ------------------------------------------------------------------------------------------ frozen linears
# The frozen decoder layers, lm_head and the encoders only ever need y = act(x W^T + b) and dx = dy W.  Both run on
# mmgl_gemm_nt (the persistent ping-pong MFMA kernel of csrc/gemm8p.hip for large bf16 shapes): forward with the bias /
# activation epilogue, dgrad as an NT GEMM against a cached W^T copy (the weights never change), and fc1's ReLU backward
# folded into the epilogue of fc2's dgrad (zmask) -- no clamp pass, no pre-activation kept, no library GEMM.
_WT_CACHE = {}
ACT_CODES = {"none": 0, "relu": 1, "gelu": 2, "quick_gelu": 3, "gelu_new": 4, "gelu_pytorch_tanh": 4, "gelu_fast": 4}


def _transposed(weight, pad_to=1):
    """W^T as a contiguous tensor [in, out_padded] (out zero-padded up to a multiple of `pad_to`), cached per weight OBJECT
    (weakref-checked: an id or an address can be reused by a later tensor) and rebuilt when its storage / version / dtype
    changes (load_state_dict, .bfloat16(), flattening)."""
    key = (id(weight), pad_to)
    tag = (weight.data_ptr(), weight._version, weight.dtype, tuple(weight.shape))
    hit = _WT_CACHE.get(key)
    if hit is None or hit[0]() is not weight or hit[1] != tag:
        if len(_WT_CACHE) > 4096:                             # drop entries of dead tensors
            for k in [k for k, h in _WT_CACHE.items() if h[0]() is None]:
                del _WT_CACHE[k]
        with torch.no_grad():
            wt = weight.detach().t()
            n = wt.shape[1]
            npad = (n + pad_to - 1) // pad_to * pad_to
            if npad != n:
                full = wt.new_zeros(wt.shape[0], npad)
                full[:, :n] = wt
                wt = full
            hit = (weakref.ref(weight), tag, wt.contiguous())
        _WT_CACHE[key] = hit
    return hit[2]


_tile_counters = {}


def gemm_dynamic_schedule(enable=True, device=None):
    """Switch this process's persistent-GEMM launches between the static tile schedule (default) and the dynamic one
    (mmgl_gemm_set_tile_counter): workgroups take tiles from per-XCD counters as they become free.  The data-parallel engine
    turns it on when it runs with more than one rank: the bucket all-reduces of the backward pass share the CUs with these GEMMs,
    and a statically scheduled GEMM waits for its displaced workgroups (reference DDP overlap: run_generation.py:317-319, 485).
    One counter block per device, process-wide (the backward GEMMs are launched from autograd's device thread); every launch leaves it
    zeroed, GEMMs of one device run on one stream."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):                          # the C side keys the setting by hipGetDevice
        if not enable:
            lib().mmgl_gemm_set_tile_counter(None)
            return None
        c = _tile_counters.get(device)
        if c is None:
            c = _tile_counters[device] = torch.zeros(16, dtype=torch.int32, device=device)
        lib().mmgl_gemm_set_tile_counter(ptr(c))
    return c


def gemm_nt(x2, w, bias=None, residual=None, zmask=None, act=0, out_scale=1.0, K=None, out=None):
    """Raw mmgl_gemm_nt call: act((x2[:, :K] @ w[:, :K]^T + bias) * out_scale) [zeroed where zmask <= 0] [+ residual].
    x2 [M, >=K] and w [N, >=K] are row-major with unit column stride (their row strides are passed as ldx / ldw: K may exceed
    x2's row length when the matching columns of w are zero padding).  No autograd."""
    require_cuda(x2, w)
    M, N = x2.shape[0], w.shape[0]
    K = x2.shape[1] if K is None else K
    if x2.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError("gemm_nt: operands need unit column stride")
    y = torch.empty(M, N, dtype=x2.dtype, device=x2.device) if out is None else out
    if M == 0:
        return y
    for name, t in (("zmask", zmask), ("residual", residual)):
        if t is not None and (t.stride(1) != 1 or t.stride(0) != y.stride(0)):
            raise ValueError(f"gemm_nt: {name} must share the output's row stride ({t.stride(0)} vs {y.stride(0)})")
    nws = lib().mmgl_gemm_nt_workspace(M, N, K, x2.stride(0), w.stride(0), y.stride(0), dtype_code(x2))
    ws = torch.empty(nws, dtype=torch.uint8, device=x2.device) if nws else None       # K-split partial tiles (few-tile shapes)
    _lib.call("mmgl_gemm_nt", dict(flops=2.0 * M * N * K, bytes=float(M * K + N * K + M * N) * x2.element_size(),
                                   tag=f"{M}x{N}x{K}" + ("+b" if bias is not None else "") + ("+z" if zmask is not None else "")
                                       + ("+r" if residual is not None else "") + (f"+a{act}" if act else "")),
              ptr(x2), x2.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(residual), ptr(zmask), ptr(y), y.stride(0), M, N, K, act,
              float(out_scale), ptr(ws), nws, dtype_code(x2), stream_ptr())
    return y


def relu_bits_bytes(M, N, K, ldx, ldw, ldy, dtype):
    """Bytes of the ReLU mask bits of an [M, N] = relu(x W^T) output (0: that shape does not run as whole tiles of the persistent
    kernel, or the dtype is not bf16 -> keep the activation as the mask)."""
    if dtype != torch.bfloat16:
        return 0
    return lib().mmgl_gemm_nt_relu_bits_bytes(M, N, K, ldx, ldw, ldy, _lib.BF16)


def gemm_nt_relu_bits(x2, w, bias, out, K=None):
    """out = relu(x2[:, :K] @ w^T + bias) plus one bit per element (out > 0), lane-private to the persistent GEMM's tiling:
    (out, bits uint8 [nbytes]).  Callers check relu_bits_bytes() > 0 first.  No autograd."""
    M, N = x2.shape[0], w.shape[0]
    K = x2.shape[1] if K is None else K
    nb = relu_bits_bytes(M, N, K, x2.stride(0), w.stride(0), out.stride(0), x2.dtype)
    if not nb:
        raise ValueError("gemm_nt_relu_bits: shape not eligible (relu_bits_bytes == 0)")
    bits = torch.empty(nb, dtype=torch.uint8, device=x2.device)
    _lib.call("mmgl_gemm_nt_relu_bits", dict(flops=2.0 * M * N * K, bytes=float(M * K + N * K + M * N) * x2.element_size(), tag=f"{M}x{N}x{K}+b+a1+bits"),
              ptr(x2), x2.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), ptr(bits), M, N, K, 1.0, dtype_code(x2), stream_ptr())
    return out, bits


def gemm_nt_masked(x2, w, bits, out, K=None):
    """out = (x2[:, :K] @ w^T) where the element's bit is set, 0 elsewhere (bits from gemm_nt_relu_bits of the same [M, N]).  No autograd."""
    M, N = x2.shape[0], w.shape[0]
    K = x2.shape[1] if K is None else K
    _lib.call("mmgl_gemm_nt_masked", dict(flops=2.0 * M * N * K, bytes=float(M * K + N * K + M * N) * x2.element_size(), tag=f"{M}x{N}x{K}+bits"),
              ptr(x2), x2.stride(0), ptr(w), w.stride(0), ptr(bits), ptr(out), out.stride(0), M, N, K, 1.0, dtype_code(x2), stream_ptr())
    return out


def _gemm_nt_padded(x2, w, bias=None, zmask=None, act=0, K=None):
    """gemm_nt for feature counts the MFMA kernels cannot address (K not a multiple of one 16-byte chunk, N not a multiple
    of 8 -- tiny test vocabularies, the Laplacian-PE width): operands are zero-padded, the result sliced back."""
    kq = 8 if x2.dtype == torch.bfloat16 else 4
    Kx = x2.shape[1] if K is None else K
    N = w.shape[0]
    pk, pn = (-Kx) % kq, (-N) % 8
    if not pk and not pn:
        return gemm_nt(x2, w, bias, zmask=zmask, act=act, K=K)
    x2 = F.pad(x2[:, :Kx], (0, pk)) if pk else x2
    w = F.pad(w[:, :Kx], (0, pk, 0, pn))
    bias = F.pad(bias, (0, pn)) if (bias is not None and pn) else bias
    zmask = F.pad(zmask, (0, pn)) if (zmask is not None and pn) else zmask
    y = gemm_nt(x2.contiguous(), w.contiguous(), bias, zmask=zmask, act=act, K=Kx + pk)
    return y[:, :N].contiguous() if pn else y


def frozen_dgrad(g, weight, zmask=None, out=None, bits=None, out_scale=1.0, residual=None):
    """dx[M,K] = g[M,N] @ W[N,K] for a frozen W  ==  an NT GEMM against the cached W^T [K, Npad].  The contraction length is
    padded to a multiple of 128 with zero columns of W^T (lm_head: N = vocab = 50272) and g is read with its own row stride.
    No autograd."""
    N, K = weight.shape
    pad = 128 if (g.dtype == torch.bfloat16 and N % 128) else 1
    wt = _transposed(weight, pad) if weight.dtype == g.dtype else weight.detach().to(g.dtype).t().contiguous()
    kk = wt.shape[1]
    if kk != N and not lib().mmgl_gemm_nt_fast(g.shape[0], K, kk, g.stride(0), kk, K, dtype_code(g)):
        wt, kk = wt[:, :N].contiguous(), N                # shape not on the fast path: dense operands
    if out is None and zmask is not None and zmask.stride(0) != zmask.shape[1]:
        # the mask (a ReLU output kept with a padded row pitch, see _ffn_pitch) shares the output's row stride in the epilogue
        out = torch.empty(zmask.shape[0], zmask.stride(0), dtype=g.dtype, device=g.device)[:, :zmask.shape[1]]
    if bits is not None:                                      # the ReLU mask as bits (see _FrozenLinear): `out` carries the row pitch
        return gemm_nt_masked(g, wt, bits, out, K=kk)
    if out is not None and kk % (8 if g.dtype == torch.bfloat16 else 4) == 0 and K % 8 == 0:
        return gemm_nt(g, wt, zmask=zmask, residual=residual, out_scale=out_scale, K=kk, out=out)
    if residual is not None or out_scale != 1.0:
        if kk % (8 if g.dtype == torch.bfloat16 else 4) or K % 8:
            raise ValueError("frozen_dgrad: out_scale / residual need 16-byte aligned feature counts")
        return gemm_nt(g, wt, zmask=zmask, residual=residual, out_scale=out_scale, K=kk)
    dx = _gemm_nt_padded(g, wt, zmask=zmask, K=kk)
    if out is not None:
        out.copy_(dx)
        return out
    return dx




def _ffn_pitch(M, N, K, dtype):
    """Row pitch (elements) for the [M, N] hidden buffer between the two linears of a frozen FFN.  A power-of-two row size
    (8192 bf16 = 16 KiB) costs the persistent GEMM ~3 % at that shape (1049 vs 1015 us, tools/probes/gemm_pitch2.py): the buffer
    and its gradient get 128 extra columns of pitch when all four GEMMs that touch them take strided operands (fast path)."""
    if dtype != torch.bfloat16 or (N * 2) % 8192:
        return N
    P, L = N + 128, lib()
    if L.mmgl_gemm_nt_fast(M, N, K, K, K, P, _lib.BF16) == 1 and L.mmgl_gemm_nt_fast(M, K, N, P, N, K, _lib.BF16) == 1:
        return P
    return N


_CU_COUNT = {}


def _round_split_rows(M, N, device):
    """Rows [0, m1) of an [M, N] GEMM output that fill whole rounds of the persistent kernel's 256 x 256 tiles, when the rest is a
    short tail: at the reference's batch (M = 4 * 640 = 2560, reference language_modelling/run_generation.py:124-126) the FFN's
    [2560, 8192] output is 320 tiles on 256 CUs -- a second round at a quarter of the chip (102.6 us); 2048 rows on the persistent
    kernel + 512 rows on the few-tile kernel take 78.3 us (tools/probes/gemm_msplit.py).  0 = do not split."""
    G = _CU_COUNT.get(device.index)
    if G is None:
        G = _CU_COUNT[device.index] = torch.cuda.get_device_properties(device).multi_processor_count
    tm, tn = (M + 255) // 256, (N + 255) // 256
    rounds, rem = divmod(tm * tn, G)
    if not (1 <= rounds <= 3 and 0 < rem <= G // 4) or (rounds * G) % tn:
        return 0
    m1 = rounds * G // tn * 256
    return m1 if 0 < m1 < M else 0


def _row_strided(t2, cols, n_out):
    """t2 [M, cols] as a GEMM operand: itself when it is row-major with unit column stride and a row pitch the fast path
    takes, else a contiguous copy."""
    if t2.is_contiguous():
        return t2
    if (t2.stride(1) == 1 and t2.stride(0) >= cols and t2.stride(0) % 8 == 0 and t2.dtype == torch.bfloat16
            and lib().mmgl_gemm_nt_fast(t2.shape[0], n_out, cols, t2.stride(0), cols, n_out, _lib.BF16)):
        return t2
    return t2.contiguous()


# The ReLU mask bits fc1's forward produced travel from _FrozenLinear.forward to frozen_linear (its caller, a few lines below, which hangs
# them on the output tensor) through this per-THREAD slot: autograd Functions can only return tensors.  A consumer that does not find
# the attribute (a wrapper or view dropped it) falls back to the activation as the mask.
_HANDOVER = threading.local()
_HANDOVER.relu_bits = None


class _FrozenLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, mask_dx, premasked, residual=None, relu_bits=None, relu_pitch=0, relu_rows=0):
        require_cuda(x, weight)
        K, N = weight.shape[1], weight.shape[0]
        x2 = _row_strided(x.reshape(-1, K), K, N)
        w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
        b = None if bias is None else (bias if bias.dtype == x.dtype else bias.to(x.dtype))
        kq = 8 if x.dtype == torch.bfloat16 else 4
        if K % kq == 0 and N % 8 == 0:
            # the output is allocated in its final shape and returned as is (not a view made inside this Function): consumers
            # such as the in-place rotary embedding may then modify it
            pitch = _ffn_pitch(x2.shape[0], N, K, x.dtype) if (premasked and residual is None) else N
            if pitch != N:                   # fc1 of a frozen FFN: the consumer (fc2, mask_dx) reads it with its row stride
                out = torch.empty(*x.shape[:-1], pitch, dtype=x.dtype, device=x.device)[..., :N]
                wc = w.contiguous()
                M_ = x2.shape[0]
                m1 = _round_split_rows(M_, N, x.device) if act == 1 else 0
                if m1 and relu_bits_bytes(m1, N, K, x2.stride(0), K, pitch, x.dtype):
                    # a 1.25-round output: whole rounds on the persistent kernel (mask bits), the short tail on the few-tile kernel
                    # (its rows of the activation stay the mask of fc2's backward)
                    y = out.reshape(-1, N)
                    _, bits = gemm_nt_relu_bits(x2[:m1], wc, b, y[:m1])
                    gemm_nt(x2[m1:], wc, b, act=1, out=y[m1:])
                    _HANDOVER.relu_bits = (bits, pitch, m1)
                elif act == 1 and relu_bits_bytes(M_, N, K, x2.stride(0), K, pitch, x.dtype):
                    # the ReLU mask leaves as bits beside the activation: fc2's backward applies those (16 bytes per lane and
                    # tile) instead of re-reading -- and keeping -- the [M, ffn] activation
                    y, bits = gemm_nt_relu_bits(x2, wc, b, out.reshape(-1, N))
                    _HANDOVER.relu_bits = (bits, pitch, M_)
                else:
                    y = gemm_nt(x2, wc, b, act=act, out=out.reshape(-1, N))
            else:
                out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
                r2 = None if residual is None else residual.reshape(-1, N).contiguous()
                y = gemm_nt(x2, w.contiguous(), b, residual=r2, act=act, out=out.view(-1, N))
        else:
            out = None
            y = _gemm_nt_padded(x2, w.contiguous(), b, act=act)
            if residual is not None:
                y = y + residual.reshape(-1, N)
        ctx.has_resid = residual is not None
        # premasked: the consumer folds this layer's ReLU backward into its own dgrad (mask_dx there): differentiate as a plain
        # linear and keep nothing.  mask_dx: x2 is a ReLU output whose backward rides in this layer's dgrad epilogue -- as the
        # producer's mask bits when it left some and the dgrad's shape takes them (then x2 itself is not kept), else as x2.
        ctx.mask_bits = None
        xkeep = x2 if mask_dx else None
        if mask_dx and relu_bits is not None:
            mrows = relu_rows or x2.shape[0]                  # rows the bits cover (the rest: the activation rows themselves)
            if x2.dtype == torch.bfloat16 and x2.stride(0) == relu_pitch and N % 128 == 0 and 0 < mrows <= x2.shape[0] and \
                    relu_bits_bytes(mrows, K, N, N, N, relu_pitch, x2.dtype) == relu_bits.numel():
                ctx.mask_bits = (relu_bits, relu_pitch, mrows)
                xkeep = None
                if mrows < x2.shape[0]:      # a short tail, copied (a view would keep the [M, ffn] buffer) with the row pitch the dgrad's output has
                    xkeep = torch.empty(x2.shape[0] - mrows, relu_pitch, dtype=x2.dtype, device=x2.device)[:, :K]
                    xkeep.copy_(x2[mrows:])
        ctx.save_for_backward(weight, y if (act == 1 and not premasked) else None, xkeep)
        ctx.act = 0 if premasked else act
        ctx.xshape = x.shape
        return out if out is not None else y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        weight, y, xmask = ctx.saved_tensors
        N, K = weight.shape
        g = dy.reshape(-1, N)
        g = _row_strided(g, N, K) if ctx.act == 0 else g.contiguous()
        if ctx.act == 1:
            gm = torch.empty_like(g)
            _lib.call("mmgl_relu_bwd", dict(bytes=3.0 * g.numel() * g.element_size()), ptr(g), ptr(y), ptr(gm), g.numel(), dtype_code(g), stream_ptr())
            g = gm
        elif ctx.act:
            raise RuntimeError("frozen_linear: only the ReLU epilogue is differentiable")
        if ctx.mask_bits is not None:
            bits, pitch, mrows = ctx.mask_bits
            buf = torch.empty(g.shape[0], pitch, dtype=g.dtype, device=g.device)[:, :K]
            if mrows < g.shape[0]:                            # whole rounds with the mask bits, the tail rows with their activation rows
                frozen_dgrad(g[:mrows], weight, out=buf[:mrows], bits=bits)
                frozen_dgrad(g[mrows:], weight, zmask=xmask, out=buf[mrows:])
                dx = buf
            else:
                dx = frozen_dgrad(g, weight, out=buf, bits=bits)
        else:
            dx = frozen_dgrad(g, weight, zmask=xmask)
        return dx.reshape(ctx.xshape), None, None, None, None, None, (dy if ctx.has_resid else None), None, None, None


def frozen_linear(x, weight, bias, relu=False, mask_dx=False, bwd_premasked=False, act=None, residual=None):
    """act(x W^T + b) for a FROZEN nn.Linear (reference :194-199, :273, :352-355 inside the frozen LM layers, lm_head :826).
    mask_dx / bwd_premasked: as in `linear` -- `h = frozen_linear(x, W1, b1, relu=True, bwd_premasked=True);
    y = frozen_linear(h, W2, b2, mask_dx=True)` puts fc1's ReLU backward into the epilogue of fc2's dgrad GEMM.
    residual: a differentiable [..., out_features] tensor added in the GEMM epilogue (the LoRA update of an adapted projection)."""
    if weight.requires_grad or (bias is not None and bias.requires_grad):
        raise ValueError("frozen_linear: weight and bias must be frozen (requires_grad=False)")
    if x.shape[-1] != weight.shape[1]:
        raise ValueError(f"frozen_linear: x has {x.shape[-1]} features, weight expects {weight.shape[1]}")
    code = 1 if relu else ACT_CODES.get(act or "none")
    if code is None:
        raise ValueError(f"frozen_linear: activation {act!r} is not fused")
    if code > 1 and torch.is_grad_enabled() and x.requires_grad:
        raise ValueError("frozen_linear: GELU epilogues are forward-only (frozen encoders); use relu / none under autograd")
    if bwd_premasked and code != 1:
        raise ValueError("frozen_linear: bwd_premasked needs the ReLU epilogue")
    if residual is not None and (code or residual.shape[:-1] != x.shape[:-1] or residual.shape[-1] != weight.shape[0]):
        raise ValueError("frozen_linear: `residual` ([..., out_features], added in the GEMM epilogue) goes with no activation")
    _HANDOVER.relu_bits = None
    mb = getattr(x, "_mmgl_relu_bits", None) if mask_dx else None
    y = _FrozenLinear.apply(x, weight, bias, code, bool(mask_dx), bool(bwd_premasked), residual, None if mb is None else mb[0], 0 if mb is None else mb[1],
                            0 if mb is None else mb[2])
    if _HANDOVER.relu_bits is not None:                      # fc1 of a frozen FFN left its ReLU mask as bits: hand them to the consumer
        y._mmgl_relu_bits, _HANDOVER.relu_bits = _HANDOVER.relu_bits, None
    return y


def frozen_linear_relu(x, weight, bias):
    return frozen_linear(x, weight, bias, relu=True)


# ------------------------------------------------------------------------------------------ Llama-family elementwise ops
class _RopeQK(torch.autograd.Function):
    """Rotary embedding of the q and k blocks of a fused-QKV buffer [B, T, 3*H*D], in place (the buffer is a fresh GEMM output)."""

    @staticmethod
    def forward(ctx, qkv, cos_sin, num_heads):
        require_cuda(qkv, cos_sin)
        B, T, d3 = qkv.shape
        D = d3 // 3 // num_heads
        if not qkv.is_contiguous():
            raise ValueError("rope_qk_: qkv must be contiguous")
        _lib.call("mmgl_rope_inplace", dict(bytes=2.0 * B * T * (2 * d3 // 3) * qkv.element_size()), ptr(qkv), ptr(cos_sin), B * T, T, num_heads, D, d3, 2, 0,
                  dtype_code(qkv), stream_ptr())
        ctx.mark_dirty(qkv)
        ctx.save_for_backward(cos_sin)
        ctx.meta = (T, num_heads, D)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        (cos_sin,) = ctx.saved_tensors
        T, H, D = ctx.meta
        dqkv = dqkv.contiguous()                                         # autograd owns the incoming buffer: rotate INTO a new one, never in place
        out = torch.empty_like(dqkv)
        rows = dqkv.numel() // dqkv.shape[-1]
        _lib.call("mmgl_rope", dict(bytes=2.0 * rows * dqkv.shape[-1] * dqkv.element_size()), ptr(dqkv), ptr(out), ptr(cos_sin), rows, T, H, D,
                  dqkv.shape[-1], 2, 3, 1, dtype_code(dqkv), stream_ptr())
        return out, None, None


def rope_qk_(qkv, cos_sin, num_heads):
    """In-place rotary position embedding of the q and k thirds of qkv [B, T, 3*H*D] (transformers' rotate_half convention,
    position = index along T).  cos_sin: fp32 [T, D/2, 2]."""
    if qkv.dim() != 3 or qkv.shape[2] % (3 * num_heads):
        raise ValueError(f"rope_qk_: qkv{tuple(qkv.shape)} is not [B, T, 3*H*D] for H={num_heads}")
    D = qkv.shape[2] // 3 // num_heads
    if cos_sin.dtype != torch.float32 or tuple(cos_sin.shape) != (qkv.shape[1], D // 2, 2):
        raise ValueError(f"rope_qk_: cos_sin must be fp32 [T={qkv.shape[1]}, D/2={D // 2}, 2], got {cos_sin.dtype} {tuple(cos_sin.shape)}")
    return _RopeQK.apply(qkv, cos_sin.contiguous(), num_heads)


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gu):
        require_cuda(gu)
        F2 = gu.shape[-1]
        g2 = gu.contiguous().view(-1, F2)
        y = torch.empty(g2.shape[0], F2 // 2, dtype=gu.dtype, device=gu.device)
        _lib.call("mmgl_swiglu_fwd", dict(bytes=1.5 * g2.numel() * g2.element_size()), ptr(g2), ptr(y), g2.shape[0], F2 // 2, dtype_code(g2), stream_ptr())
        ctx.save_for_backward(g2)
        ctx.shape = gu.shape
        return y.view(*gu.shape[:-1], F2 // 2)

    @staticmethod
    def backward(ctx, dy):
        (g2,) = ctx.saved_tensors
        M, F2 = g2.shape
        dy2 = dy.contiguous().view(M, F2 // 2)
        dgu = torch.empty_like(g2)
        _lib.call("mmgl_swiglu_bwd", dict(bytes=2.5 * g2.numel() * g2.element_size()), ptr(dy2), ptr(g2), ptr(dgu), M, F2 // 2, dtype_code(g2), stream_ptr())
        return dgu.view(ctx.shape)


def swiglu(gate_up):
    """silu(gate_up[..., :F]) * gate_up[..., F:] over one fused [gate | up] projection output (LlamaMLP)."""
    if gate_up.shape[-1] % 2:
        raise ValueError("swiglu: last dim must be 2*F")
    return _SwiGLU.apply(gate_up)


# ------------------------------------------------------------------------------------------ frozen encoders (forward only)
def encoder_attention(q, k, v, cu_seqlens, num_heads, max_len, q_rows=None, work=None):
    """Bidirectional attention over packed sequences (no padding rows exist).  q, k, v: [ntok, H*D] views with a common
    row stride (e.g. the three column slices of a fused-QKV output); q pre-scaled by D^-1/2; cu_seqlens int32 [nseq+1].
    Returns [ntok, H*D]; with q_rows < max_len only the first q_rows rows of every sequence are written.
    No autograd: the encoders are frozen (reference modelling_cross_attention.py:922-934)."""
    require_cuda(q, k, v, cu_seqlens)
    ntok, hd = q.shape
    if k.shape != q.shape or v.shape != q.shape or hd % num_heads:
        raise ValueError(f"encoder_attention: incompatible shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)} H={num_heads}")
    ld = q.stride(0)
    if q.stride(1) != 1 or k.stride() != q.stride() or v.stride() != q.stride():
        raise ValueError("encoder_attention: q, k, v need unit column stride and one common row stride")
    if cu_seqlens.dtype != torch.int32:
        raise ValueError("encoder_attention: cu_seqlens must be int32")
    out = torch.empty(ntok, hd, dtype=q.dtype, device=q.device)
    nseq = cu_seqlens.numel() - 1
    if nseq <= 0 or ntok == 0:
        return out
    _lib.call("mmgl_encattn_fwd", work, ptr(q), ptr(k), ptr(v), ptr(cu_seqlens), ptr(out), nseq, num_heads, hd // num_heads, ld, hd,
              int(max_len), int(max_len if q_rows is None else q_rows), dtype_code(q), stream_ptr())
    return out


def add_layer_norm(x, res, gamma, beta, eps, return_sum=False):
    """y = LayerNorm(x + res) in one pass (optionally also returning the sum = the new residual stream).  Forward only."""
    require_cuda(x, res)
    if x.shape != res.shape:
        raise ValueError(f"add_layer_norm: shapes differ {tuple(x.shape)} vs {tuple(res.shape)}")
    cols = x.shape[-1]
    x2, r2 = x.contiguous().view(-1, cols), res.contiguous().view(-1, cols)
    y = torch.empty_like(x2)
    s = torch.empty_like(x2) if return_sum else None
    g, b = gamma.to(x.dtype).contiguous(), beta.to(x.dtype).contiguous()
    if x2.shape[0]:
        _lib.call("mmgl_add_layernorm_fwd", dict(bytes=(4.0 if return_sum else 3.0) * x2.numel() * x2.element_size()), ptr(x2), ptr(r2), ptr(g), ptr(b), ptr(s), ptr(y), None, None, x2.shape[0], cols, float(eps),
                  0.0, 0, dtype_code(x2), stream_ptr())
    return (s.view(x.shape), y.view(x.shape)) if return_sum else y.view(x.shape)


def activation_(x, name):
    """In-place HF ACT2FN[name] over a contiguous tensor (one pass).  Forward only."""
    require_cuda(x)
    if name not in ACT_CODES:
        raise ValueError(f"activation_: unsupported activation {name!r}")
    if not x.is_contiguous():
        raise ValueError("activation_: tensor must be contiguous")
    _lib.call("mmgl_activation_fwd", dict(bytes=2.0 * x.numel() * x.element_size()), ptr(x), ptr(x), x.numel(), ACT_CODES[name], dtype_code(x), stream_ptr())
    return x
