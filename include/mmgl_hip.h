/*
 * mmgl_hip.h -- C ABI of libmmgl_hip.so: the MI355X (gfx950) kernels behind MMGL's
 * neighbor-fusion hot path.
 *
 * The reference (minjiyoon/MMGL) has no FFI: the path sits behind a Python module API
 * (SURVEY.md 8b).  Each entry point below replaces a run of stock ATen ops inside one reference
 * function; the citation after "replaces:" is reference file:line.  The Python binding that a
 * maintainer of the reference would add is in INTEGRATION.md (ctypes, one torch.autograd.Function
 * per fwd/bwd pair).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated inside; scratch
 *     is passed explicitly (query the size with the matching *_workspace function).
 *   - tensors are dense row-major in the layout given in the comment.
 *   - dtype: element type of the activations (MMGL_F32 / MMGL_BF16); reductions, softmax and
 *     accumulation are always fp32.
 *   - last argument: the hipStream_t to launch on, passed as void* (0 = default stream).
 *   - returns 0 on success, otherwise an MMGL_ERR_* code; mmgl_last_error() gives the message.
 *     The Python shim maps INVALID/UNSUPPORTED to ValueError (the reference raises ValueError for
 *     shape / mode mismatches, modelling_cross_attention.py:160-164,214-224,260-264) and HIP to
 *     RuntimeError.
 *   - re-entrant.  Global state: the last-error string (thread-local), and ONE opt-in per-device setting, the dynamic tile
 *     schedule of the persistent GEMM (mmgl_gemm_set_tile_counter below): while it is set, that device's persistent-GEMM launches
 *     COUNTERS are bound to one stream (the first one that launches after the call); a launch on another stream runs on the static
 *     schedule (MMGL_GEMM_STRICT_STREAM=1: returns MMGL_ERR_INVALID instead).  With the default static schedule nothing is shared
 *     between calls.
 */
#ifndef MMGL_HIP_H
#define MMGL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MMGL_F32 = 0, MMGL_BF16 = 1 };
enum { MMGL_OK = 0, MMGL_ERR_INVALID = 1, MMGL_ERR_UNSUPPORTED = 2, MMGL_ERR_HIP = 3 };
enum { MMGL_ACT_NONE = 0, MMGL_ACT_RELU = 1 };

const char* mmgl_last_error(void);
int mmgl_version(void);

/* ---------------------------------------------------------------------------------------------
 * Masked cross-attention core:  O = softmax(max(Q K^T + M, finfo.min)) V   per (batch, head)
 * replaces: MPTAttention.forward, model/modelling_cross_attention.py:206-271 (head split, bmm,
 *           additive mask + clamp, softmax, bmm, head merge) and _expand_mask :68-79 (the
 *           [B,1,T,S] additive mask is never materialised: the [B,S] byte mask is consumed).
 *   q         [B,T,H*D]  already projected AND scaled by D^-0.5 (:194)
 *   k, v      [B,S,H*D]  projected neighbor tokens (:198-199)
 *   key_valid [B,S] uint8, 1 = attend (neighbor pos_id > 0, :1077/:1084-1104)
 *   out       [B,T,H*D]
 *   lse       [B,H,T] fp32 log-sum-exp of the masked scores (saved for backward)
 * A sample with no valid key yields the uniform distribution over its S keys (the reference's
 * finfo.min clamp, :226-228), never NaN.  D in {16,32,64,128}; S <= 256.
 * No attention dropout here (:256): OPT's attention_dropout is 0.0, where it is the identity; a config that asks for it (or for a
 * head mask / output_attentions) is routed to mmgl_attn_general_* below by the host mirror.
 */
int mmgl_xattn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid,
                   void* out, float* lse, int B, int H, int T, int S, int D, int dtype, void* stream);

/* Backward of the above.  dq [B,T,H*D], dk/dv [B,S,H*D].  `workspace` must hold
 * mmgl_xattn_bwd_workspace(...) bytes (fp32 row-dots + per-chunk dK/dV partials, reduced in a
 * fixed order => bitwise deterministic).  On a sample with no valid key dQ/dK are halved exactly
 * as autograd does for the reference's torch.max tie (see oracle/lm_ref.py attention_core). */
size_t mmgl_xattn_bwd_workspace(int B, int H, int T, int S, int D);
int mmgl_xattn_bwd(const void* dout, const void* q, const void* k, const void* v, const float* lse,
                   const uint8_t* key_valid, void* dq, void* dk, void* dv,
                   void* workspace, size_t workspace_bytes,
                   int B, int H, int T, int S, int D, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Causal self-attention of the (frozen) decoder layers, flash style:  O = softmax(mask(Q K^T)) V,
 * mask = (s <= t) AND key_valid[b,s].
 * replaces: MPTAttention.forward self branch, model/modelling_cross_attention.py:203-271 with the additive masks of
 *           _make_causal_mask / _expand_mask (:51-79, :455-476): neither the [B,1,T,T] mask nor the [B,H,T,T] scores exist.
 *   q [B,T,H*D] already scaled by D^-0.5 ; k, v [B,T,H*D] ; key_valid [B,T] uint8 (the LM attention_mask)
 *   out [B,T,H*D] ; lse [B,H,T] fp32.
 * Precondition: key_valid[b,0] == 1 for every b (right-padded sequences, wikiweb2m/data.py:321-333), so every query row
 * keeps at least one key and the result equals the reference's finfo.min-clamped softmax; the host wrapper checks it.
 * SURVEY.md 8(f) row 2.  Backward returns dq, dk, dv (the layers are frozen, activations still need gradients);
 * `out` is the forward output (delta = rowsum(dO*O)); workspace >= mmgl_selfattn_bwd_workspace bytes.
 * ld_qkv / ld_dqkv: row stride in elements of q,k,v / dq,dk,dv; 0 = packed (H*D).  With 3*H*D the three pointers are
 * column slices of ONE fused projection output / gradient buffer (one fused-QKV GEMM forward, one dgrad GEMM backward).
 * out, dout are always packed.
 */
int mmgl_selfattn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out, float* lse,
                      int B, int H, int T, int D, int ld_qkv, int dtype, void* stream);
size_t mmgl_selfattn_bwd_workspace(int B, int H, int T);
int mmgl_selfattn_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
                      const uint8_t* key_valid, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                      int B, int H, int T, int D, int ld_qkv, int ld_dqkv, int dtype, void* stream);
/* The same attention with P always-visible PREFIX keys in front of the T causal ones: key s is visible to query t iff
 * s <= t + P (and key_valid[b][s]).
 * replaces: the self-attention of an OPT layer under peft prefix tuning (reference model/modelling_self_attention.py:88-93,
 *   PrefixTuningConfig(num_virtual_tokens=20): a learned per-layer key/value prefix that HF concatenates in front of the layer's
 *   keys and values as past_key_values).
 *   q, dq [B,T,·] (row strides ld_q / ld_dq), k, v, dk, dv [B,P+T,·] (ld_kv / ld_dkv), key_valid [B,P+T], out, dout [B,T,H*D]
 *   packed, lse [B,H,T].  P = 0 is mmgl_selfattn_fwd / _bwd.  Workspace: mmgl_selfattn_bwd_workspace(B, H, T). */
int mmgl_selfattn_prefix_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out, float* lse,
                             int B, int H, int T, int P, int D, int ld_q, int ld_kv, int dtype, void* stream);
int mmgl_selfattn_prefix_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
                             const uint8_t* key_valid, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                             int B, int H, int T, int P, int D, int ld_q, int ld_kv, int ld_dq, int ld_dkv, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * General attention core: the options of MPTAttention.forward the fused kernels above leave out -- attention-probability dropout,
 * layer_head_mask, output_attentions -- for both call sites of the class (causal = 0: the gated cross-attention layers, key mask only;
 * causal = 1: the decoder's self-attention, (s <= t) AND key_valid, S == T).
 * replaces: model/modelling_cross_attention.py:206-271 incl. :237-244 (layer_head_mask scales the probabilities of a head),
 *           :246-254 (attn_weights_reshaped: the head-masked probabilities, returned BEFORE dropout), :256 (nn.functional.dropout on
 *           the probabilities); model/modelling_self_attention.py reaches the same class through the OPT decoder.
 *   q [B,T,H*D] already scaled; k, v [B,S,H*D]; key_valid [B,S] uint8; head_mask [H] fp32 or NULL
 *   out [B,T,H*D]; probs [B,H,T,S] (dtype of q) or NULL; lse [B,H,T] fp32 (+inf marks a query row without any allowed key: uniform
 *   over all S keys, as the reference's finfo.min clamp gives)
 *   p_drop in [0, 1): probability of dropping a probability; the keep mask is the counter hash of (seed, ((b H + h) T + t) S + s)
 *   every dropout of this library uses, regenerated by the backward from the same (p_drop, seed) -- nothing is stored.
 * Written for exactness, not speed (fp32 VALU arithmetic, two passes over the keys): these options are inert on every BASELINE config
 * (attention_dropout = 0, no head masks); every other call stays on mmgl_xattn_* / mmgl_selfattn_*.  D <= 128, any S.
 * mmgl_attn_dropout_mask writes the keep mask as bytes [B,H,T,S] (1 = kept): a test / debug entry point (the parity tests hand it to
 * the oracle so both sides drop the same probabilities).
 */
int mmgl_attn_general_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, const float* head_mask, void* out,
                          void* probs, float* lse, int B, int H, int T, int S, int D, int causal, float p_drop,
                          uint64_t seed, int dtype, void* stream);
size_t mmgl_attn_general_bwd_workspace(int B, int H, int T);
int mmgl_attn_general_bwd(const void* dout, const void* q, const void* k, const void* v, const float* lse, const uint8_t* key_valid,
                          const float* head_mask, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes, int B, int H,
                          int T, int S, int D, int causal, float p_drop, uint64_t seed, int dtype, void* stream);
int mmgl_attn_dropout_mask(uint8_t* mask, int B, int H, int T, int S, float p_drop, uint64_t seed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm (affine, eps inside the sqrt) over the last dim.
 * replaces: nn.LayerNorm at modelling_cross_attention.py:319-320, 340-341, 349-350, 364-365, 635-636
 *   x,y [rows,cols]; gamma,beta [cols] (same dtype as x; may be NULL = no affine); mean,rstd [rows] fp32
 */
int mmgl_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                       int rows, int cols, float eps, int dtype, void* stream);
/* dgamma/dbeta are fp32 [cols] (may be NULL for a frozen norm); workspace >= mmgl_norm_bwd_workspace bytes. */
size_t mmgl_norm_bwd_workspace(int rows, int cols);
int mmgl_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                       void* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                       int rows, int cols, int dtype, void* stream);

/* RMSNorm  y = x * rsqrt(mean(x^2)+eps) * gamma   (Llama variant of the same block; no reference
 * counterpart in MMGL -- SURVEY.md 7.3 "config 5") */
int mmgl_rmsnorm_fwd(const void* x, const void* gamma, void* y, float* rstd,
                     int rows, int cols, float eps, int dtype, void* stream);
int mmgl_rmsnorm_bwd(const void* dy, const void* x, const void* gamma, const float* rstd,
                     void* dx, float* dgamma, void* workspace, size_t workspace_bytes,
                     int rows, int cols, int dtype, void* stream);

/* The residual add of a Llama layer fused with the RMSNorm that follows it (transformers' LlamaDecoderLayer: "hidden = residual +
 * hidden" then the next layer's input_layernorm / this layer's post_attention_layernorm; the OPT counterpart is
 * mmgl_add_layernorm_fwd/bwd):  sum_out = res + x,  y = RMSNorm(sum_out).  Backward: dres = RMSNorm'(dy) + dsum (dsum = the gradient
 * arriving on the residual stream, may be NULL) -- also the gradient of x.  dgamma fp32 optional (workspace as mmgl_rmsnorm_bwd). */
int mmgl_add_rmsnorm_fwd(const void* x, const void* res, const void* gamma, void* sum_out, void* y, float* rstd,
                         int rows, int cols, float eps, int dtype, void* stream);
int mmgl_add_rmsnorm_bwd(const void* dy, const void* dsum, const void* sum, const void* gamma, const float* rstd,
                         void* dres, float* dgamma, void* workspace, size_t workspace_bytes,
                         int rows, int cols, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gated residual:  y = residual + tanh(gate) * dropout(x, p)
 * replaces: nn.functional.dropout + "residual + tanh(gating) * h", modelling_cross_attention.py:332-335, 356-359
 *   residual,x,y [n] ; gate: device fp32 scalar (NULL => ungated, tanh(gate) := 1, :337/:361)
 *   dropout mask is a counter-based hash of (seed, element index): regenerated in backward.
 */
int mmgl_gated_residual_fwd(const void* residual, const void* x, const float* gate, void* y,
                            size_t n, float p_drop, uint64_t seed, int dtype, void* stream);
/* dres = dy (caller aliases it); dx = tanh(g)*mask*dy ; dgate (fp32 scalar, OVERWRITTEN) =
 * (1-tanh^2 g) * sum(dy * dropout(x)).  workspace >= mmgl_gated_residual_bwd_workspace(n). */
size_t mmgl_gated_residual_bwd_workspace(size_t n);
int mmgl_gated_residual_bwd(const void* dy, const void* x, const float* gate, void* dx, float* dgate,
                            void* workspace, size_t workspace_bytes,
                            size_t n, float p_drop, uint64_t seed, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Linear with fused epilogue:  y[M,N] = act( (x[M,K] @ W[N,K]^T + bias[N]) * out_scale )
 * replaces: nn.Linear q/k/v/out_proj (:194,198-199,273; out_scale = D^-0.5 for q_proj), fc1+ReLU, fc2 (:352-355)
 *   W is torch's nn.Linear layout [out_features, in_features]; bias may be NULL.
 * MFMA: bf16 -> v_mfma_f32_16x16x32_bf16 ; f32 -> v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).
 */
int mmgl_linear_fwd(const void* x, const void* W, const void* bias, void* y,
                    int M, int N, int K, int act, float out_scale, int dtype, void* stream);
/* dx[M,K] = dyp[M,N] @ W[N,K]   where dyp = dy * out_scale * act'(y)   (y = forward OUTPUT; NULL if act none).
 * workspace (>= mmgl_linear_dgrad_workspace bytes) holds W^T and, with an activation, dyp. N must be a multiple of 8. */
size_t mmgl_linear_dgrad_workspace(int M, int N, int K, int act, int dtype);
int mmgl_linear_dgrad(const void* dy, const void* y, const void* W, void* dx, void* workspace, size_t workspace_bytes,
                      int M, int N, int K, int act, float out_scale, int dtype, void* stream);
/* dW[N,K] (+)= dyp^T @ x ; dbias[N] (+)= colsum(dyp) ; both in the activation dtype; accumulate!=0 adds to
 * the existing contents (gradient accumulation across micro-batches). dbias may be NULL.
 * workspace (>= mmgl_linear_wgrad_workspace bytes) holds dyp^T and x^T. */
size_t mmgl_linear_wgrad_workspace(int M, int N, int K, int dtype);
int mmgl_linear_wgrad(const void* dy, const void* y, const void* x, void* dW, void* dbias, void* workspace,
                      size_t workspace_bytes, int M, int N, int K, int act, float out_scale, int accumulate,
                      int dtype, void* stream);
/* Whole backward of one linear in a single call: dyp is formed once, then dx / dW / dbias (each may be NULL).
 * mask_dx != 0: x is itself the output of a ReLU (fc2 after fc1+ReLU, modelling_cross_attention.py:352-355) and that ReLU's
 * backward is folded into this call: dx is zeroed where x <= 0 (in the dgrad GEMM's epilogue for the large-shape kernel), so
 * the producing linear can be differentiated with act = none on the already-masked gradient. */
size_t mmgl_linear_bwd_workspace(int M, int N, int K, int act, int dtype);
int mmgl_linear_bwd(const void* dy, const void* y, const void* x, const void* W, void* dx, void* dW, void* dbias,
                    void* workspace, size_t workspace_bytes, int M, int N, int K, int act, float out_scale,
                    int accumulate, int mask_dx, int dtype, void* stream);
/* out[C,ld] = in[R,C]^T, ld = R rounded up to a whole 16-byte chunk (zero padded) */
int mmgl_transpose(const void* in, void* out, int R, int C, int dtype, void* stream);

/* LoRA-fused linear:  y = x W^T + bias + scale * (x A^T) B^T        A [r,K], Bm [N,r]
 * (peft semantics, lora_dropout = 0; replaces peft's LoRA Linear injected at
 *  model/modelling_self_attention.py:80-87 -- third-party, parity unpinned)
 *   xa [M,r] scratch/output in the activation dtype (saved for backward). */
int mmgl_lora_linear_fwd(const void* x, const void* W, const void* bias, const void* A, const void* Bm,
                         void* y, void* xa, int M, int N, int K, int r, float scale, int dtype, void* stream);
/* dx = dy W + scale * (dy Bm) A ; dA = scale * (dy Bm)^T x ; dB = scale * dy^T xa.  dyb [M,r] scratch/output. */
size_t mmgl_lora_linear_bwd_workspace(int M, int N, int K, int r, int dtype);
int mmgl_lora_linear_bwd(const void* dy, const void* x, const void* xa, const void* W, const void* A,
                         const void* Bm, void* dx, void* dA, void* dB, void* dyb, void* workspace,
                         size_t workspace_bytes, int M, int N, int K, int r, float scale, int accumulate,
                         int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Neighbor interleave: scatter text / image neighbor tokens into slot order + build the key mask.
 * replaces: CrossAttentionModel.forward, modelling_cross_attention.py:1084-1104 (zeros + index_put x4)
 *   text_emb [B,Nt,n_tok*d], vis_emb [B,Ni,n_tok*d] (vis_emb may be NULL with Ni = 0: text_only :1075-1078)
 *   text_loc/img_loc, text_pos/img_pos: int64 [B,Nt] / [B,Ni]
 *   out_emb [B,(Nt+Ni)*n_tok,d] ; out_valid [B,(Nt+Ni)*n_tok] uint8.  Slots nobody writes stay zero/invalid.
 */
int mmgl_neighbor_interleave_fwd(const void* text_emb, const void* vis_emb, const int64_t* text_loc,
                                 const int64_t* img_loc, const int64_t* text_pos, const int64_t* img_pos,
                                 void* out_emb, uint8_t* out_valid, int B, int Nt, int Ni, int n_tok, int d,
                                 int dtype, void* stream);
int mmgl_neighbor_interleave_bwd(const void* d_out_emb, const int64_t* text_loc, const int64_t* img_loc,
                                 void* d_text_emb, void* d_vis_emb, int B, int Nt, int Ni, int n_tok, int d,
                                 int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Shifted token cross-entropy over [rows, V] logits (mean over rows whose label != ignore_index).
 * replaces: CrossEntropyLoss at modelling_cross_attention.py:831-836 (the shift is done by the caller's
 * pointer arithmetic: row r of `logits` is scored against labels[r]).
 *   loss_sum, count: fp32 device scalars (OVERWRITTEN); row_lse, row_loss [rows] fp32 (row_lse saved for backward).
 *   bwd writes dlogits = (softmax - onehot) * (*dloss_scale) / count, zero on ignored rows.
 */
int mmgl_cross_entropy_fwd(const void* logits, const int64_t* labels, float* row_lse, float* row_loss,
                           float* loss_sum, float* count, int rows, int V, int64_t ignore_index, int dtype,
                           void* stream);
int mmgl_cross_entropy_bwd(const void* logits, const int64_t* labels, const float* row_lse,
                           const float* count, const float* dloss, void* dlogits,
                           int rows, int V, int64_t ignore_index, int dtype, void* stream);

/* Learned-position ids: pos[b,t] = cumsum(mask)[b,t]*mask[b,t] - 1 + 2
 * replaces: MPTLearnedPositionalEmbedding.forward, modelling_cross_attention.py:135-145 */
int mmgl_position_ids(const int64_t* attention_mask, int64_t* pos, int B, int T, void* stream);

/* Fused AdamW step over a flat parameter bucket (torch.optim.AdamW semantics, run_generation.py:327-330).
 * param/grad in `dtype`; exp_avg/exp_avg_sq fp32; master (fp32 copy of param) may be NULL. grad is read * grad_scale. */
int mmgl_adamw_step(void* param, float* master, const void* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float grad_scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Frozen neighbor encoders, forward only (padding-free RoBERTa / CLIP-ViT passes).
 * replaces: the HF encoder calls in get_text_embs / get_visual_embs, modelling_cross_attention.py:978-1027
 * (self.text_model(...), self.visual_model(...)): their attention, residual-add + LayerNorm and activation steps.
 *
 * mmgl_encattn_fwd: bidirectional softmax(QK^T)V over PACKED sequences.  Sequence i owns rows
 *   cu_seqlens[i] .. cu_seqlens[i+1]-1 (int32 device array, nseq+1 entries) of q/k/v [ntok, ld_in] (three column slices
 *   of a fused-QKV GEMM output are fine: ld_in = 3*H*D) and of out [ntok, ld_out]; head h uses columns h*D..h*D+D-1.
 *   q must be pre-scaled by D^-1/2.  max_len = longest sequence (host knows it from the packing); q_rows = number of
 *   leading query rows per sequence to compute (pass max_len for all, 1 for the CLS row only).  No padding token is read.
 * mmgl_add_layernorm_fwd: s = x + res (rounded to dtype), y = LayerNorm(s); sum_out (may be NULL) receives s; mean, rstd
 *   [rows] fp32 may be NULL (inference).  Also the residual-add + LayerNorm pair of the decoder layers
 *   (modelling_cross_attention.py:334-350: `hidden = residual + h` followed by the next LayerNorm) when they are saved.
 *   p_drop > 0: s = res + dropout(x) (inverted dropout, the counter hash of mmgl_gated_residual_fwd under `seed`).
 * mmgl_add_layernorm_bwd: given dy (grad of y) and dsum (grad arriving on s itself, may be NULL),
 *   dres = LayerNorm'(dy) + dsum; with p_drop == 0 that is also the gradient of x (dx ignored, may be NULL), otherwise
 *   dx = dres * keep / (1 - p).  dgamma/dbeta fp32 optional (workspace as for mmgl_layernorm_bwd).
 * mmgl_activation_fwd: y = act(x) elementwise, in place allowed; act 1 relu, 2 gelu (erf), 3 quick_gelu, 4 gelu (tanh).
 */
/* General NT GEMM with fused epilogue (the frozen path's linears, their dgrads, lm_head, the encoder linears):
 *   y[M,N] = act( (x[M,K] @ W[N,K]^T + bias[N]) * out_scale )   then   y = (zmask > 0 ? y : 0)   then   y += residual
 * replaces: nn.Linear inside the frozen MPTDecoderLayer / MPTAttention (model/modelling_cross_attention.py:194-199, :273,
 *           :352-355), lm_head (:826), the RobertaModel / CLIPVisionModel linears behind :992 / :1018, and autograd's
 *           dgrad of each of them (dx = dy @ W  ==  an NT GEMM against the cached W^T; zmask = the ReLU output whose
 *           backward is folded into the epilogue).
 *   bias [N], residual [M,N], zmask [M,N] may be NULL; act: 0 none, 1 relu, 2 gelu (erf), 3 quick_gelu, 4 gelu (tanh).
 * bf16 shapes with K % 128 == 0, K >= 256, N % 16 == 0 and enough 256x256 tiles run on the persistent ping-pong kernel
 * (gemm8p.hip; mmgl_gemm_nt_fast returns 1), other bf16 shapes with K % 64 == 0, N % 8 == 0 on the 128x128 kernel (gemm_mid.hip;
 * returns 2): both take strided operands and apply the whole epilogue in the kernel.  Anything else (returns 0) is composed from
 * mmgl_linear_fwd + the elementwise kernels.
 * ldx / ldw / ldy: row strides in elements (residual and zmask share ldy); the composed path needs dense operands.
 * K may exceed ldx by less than 128 when columns ldx.. of W are zero (a contraction length padded to the kernels' K step: the
 * lm_head dgrad over a 50272-entry vocabulary): the tail of row r of x then reads the head of row r + 1 (nothing is read past
 * the last row: the buffer descriptor zero-fills).  PRECONDITION of that form: x finite -- 0 * Inf = NaN, so a non-finite element
 * at the head of row r + 1 would also poison row r of the output.
 * mmgl_relu_bwd: out = dy * (y > 0), the backward of a stand-alone ReLU epilogue (in place allowed). */
int mmgl_gemm_nt_fast(int M, int N, int K, int ldx, int ldw, int ldy, int dtype);
/* workspace: shapes with fewer 256x256 output tiles than the chip has CUs and a long contraction (the reference's batch of 4:
 * M = 2560, K >= 3072), and the last partial round of tiles of a one-to-three-round output (2560 x 8192: 320 tiles on 256 CUs),
 * are cut into K-split work items whose fp32 partial tiles live in caller memory, like every other
 * scratch of this library (no allocation inside).  mmgl_gemm_nt_workspace returns the bytes that takes (0 for every other
 * shape); with workspace == NULL or fewer bytes the same kernel runs unsplit (same result up to fp32 summation order). */
size_t mmgl_gemm_nt_workspace(int M, int N, int K, int ldx, int ldw, int ldy, int dtype);
int mmgl_gemm_nt(const void* x, int ldx, const void* W, int ldw, const void* bias, const void* residual, const void* zmask,
                 void* y, int ldy, int M, int N, int K, int act, float out_scale, void* workspace, size_t workspace_bytes,
                 int dtype, void* stream);
int mmgl_relu_bwd(const void* dy, const void* y, void* out, size_t n, int dtype, void* stream);
/* Dynamic tile schedule of the persistent bf16 GEMM kernel behind mmgl_gemm_nt / _relu_bits / _masked / mmgl_linear_*.
 * replaces: nothing in the reference -- it is what lets those GEMMs share the GPU with the gradient all-reduce that
 *   DistributedDataParallel overlaps with the backward pass (language_modelling/run_generation.py:317-319, 485): with the default
 *   static schedule (tile i * grid + workgroup) a workgroup whose CU is shared with the collective's kernel finishes its tiles
 *   late and the whole GEMM waits for it (+38 % measured); with a counter, workgroups take tiles as they become free.
 *   counter: DEVICE pointer to 16 uint32, all zero, owned by the caller and left all zero by every launch; NULL (the default)
 *   = static schedule.  The setting belongs to the CURRENT DEVICE (hipGetDevice of the caller), process-wide: it also applies to
 *   launches from other host threads -- autograd runs the backward GEMMs, the ones that overlap the collective, on its own device
 *   thread.  The counters serve ONE stream per device: the first persistent-GEMM launch after this call binds them to its stream,
 *   and a launch of that device on any other stream while they are set returns MMGL_ERR_INVALID (two launches in flight would hand
 *   out each other's tiles); setting the counter again (or NULL) releases the binding.
 *   mmgl_gemm_get_tile_counter: the current device's counter (NULL: static schedule). */
int mmgl_gemm_set_tile_counter(void* counter);
void* mmgl_gemm_get_tile_counter(void);
/* The ReLU mask of a frozen FFN as bits (replaces: keeping relu(fc1(x)) [M, ffn] for autograd's threshold_backward of
 * model/modelling_cross_attention.py:352-355, and re-reading it in fc2's dgrad).
 *   mmgl_gemm_nt_relu_bits: y = relu((x W^T + bias) * out_scale) as mmgl_gemm_nt with act = 1, plus bits_out: one bit per
 *     output element (y > 0), mmgl_gemm_nt_relu_bits_bytes() bytes of caller memory (8 KiB per 256 x 256 output tile).
 *   mmgl_gemm_nt_masked: y = (x W^T) * out_scale where the bit of that element is set, 0 elsewhere -- fc2's dgrad with fc1's
 *     ReLU backward folded in, reading 16 bytes of mask per lane and tile instead of the [M, ffn] activation.
 * The bits are private to the persistent kernel's tile -> lane mapping: write and apply them with the same M and N.
 * mmgl_gemm_nt_relu_bits_bytes returns 0 for shapes / dtypes that do not run as whole tiles of that kernel (then use
 * mmgl_gemm_nt with act = 1 and its zmask argument instead). */
size_t mmgl_gemm_nt_relu_bits_bytes(int M, int N, int K, int ldx, int ldw, int ldy, int dtype);
int mmgl_gemm_nt_relu_bits(const void* x, int ldx, const void* W, int ldw, const void* bias, void* y, int ldy, void* bits_out,
                           int M, int N, int K, float out_scale, int dtype, void* stream);
int mmgl_gemm_nt_masked(const void* x, int ldx, const void* W, int ldw, const void* bits_in, void* y, int ldy, int M, int N, int K,
                        float out_scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elementwise pieces of frozen Llama-family decoder layers (BASELINE.json config 5; the reference's fork is OPT-only,
 * model/modelling_cross_attention.py:278-375, so these follow transformers' LlamaDecoderLayer loaded through the same HF API).
 * mmgl_rope_inplace: rotary embedding of the first `nblk` column blocks (H heads x D each: q, then k) of buf [rows, ld], row r
 *   at position r % T; cos_sin [T, D/2, 2] fp32 (cos, sin); rotate_half convention; backward != 0 applies the transpose
 *   rotation (the gradient of the forward one).  In place.
 * mmgl_rope: the same from src into dst (src != dst, same [rows, ld] layout); column blocks nblk .. nall-1 (v of a fused
 *   q | k | v buffer) are copied unrotated -- the backward of the rotation without touching the gradient buffer autograd owns.
 * mmgl_swiglu_fwd: y[M,F] = silu(gate_up[:, :F]) * gate_up[:, F:]   (gate_up = one fused [gate | up] GEMM output [M, 2F]).
 * mmgl_swiglu_bwd: dgate_up[M,2F] from dy[M,F] and the saved gate_up. */
int mmgl_rope_inplace(void* buf, const float* cos_sin, size_t rows, int T, int H, int D, int ld, int nblk, int backward,
                      int dtype, void* stream);
int mmgl_rope(const void* src, void* dst, const float* cos_sin, size_t rows, int T, int H, int D, int ld, int nblk, int nall,
              int backward, int dtype, void* stream);
int mmgl_swiglu_fwd(const void* gate_up, void* y, size_t M, int F, int dtype, void* stream);
int mmgl_swiglu_bwd(const void* dy, const void* gate_up, void* dgate_up, size_t M, int F, int dtype, void* stream);

int mmgl_encattn_fwd(const void* q, const void* k, const void* v, const int32_t* cu_seqlens, void* out, int nseq,
                     int H, int D, int ld_in, int ld_out, int max_len, int q_rows, int dtype, void* stream);
int mmgl_add_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta, void* sum_out,
                           void* y, float* mean, float* rstd, int rows, int cols, float eps, float p_drop, uint64_t seed,
                           int dtype, void* stream);
int mmgl_add_layernorm_bwd(const void* dy, const void* dsum, const void* sum, const void* gamma, const float* mean,
                           const float* rstd, void* dres, void* dx, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, int rows, int cols, float p_drop, uint64_t seed, int dtype,
                           void* stream);
int mmgl_activation_fwd(const void* x, void* y, size_t n, int act, int dtype, void* stream);
/* y[i] = x[i] * (*scale), product in fp32, scale read from device memory (x == y allowed).  The upstream gradient of a scalar
 * loss -- `loss / args.grad_accumulation_steps` at reference language_modelling/run_generation.py:483 -- applied to a saved
 * gradient without a host sync. */
int mmgl_scale(const void* x, const float* scale, void* y, size_t n, int dtype, void* stream);

/* ---- gradient exchange over RCCL (xGMI) --------------------------------------------------------------------------------
 * replaces: the NCCL process group behind the reference's data-parallel wiring -- language_modelling/run_generation.py:283
 *   (dist.init_process_group("nccl")), :317-319 (DistributedDataParallel: the constructor's parameter broadcast and the bucketed
 *   gradient all-reduce of every backward pass, :485), language_modelling/utils.py:113-118 (AverageMeter.all_reduce),
 *   run_generation.py:608-616 (all_gather of the evaluation predictions).  One process per GPU; single node, like the reference.
 *   mmgl_comm_unique_id: rank 0 fills 128 bytes (ncclUniqueId) and hands them to the other ranks out of band (file, socket, env).
 *   mmgl_comm_init: every rank, with its HIP device current, joins the communicator of that id; *comm is the handle.
 *   mmgl_allreduce_sum: buf[count] <- sum over ranks, in place, on `stream` (averaging: fold 1/world into mmgl_adamw_step's
 *     grad_scale).  mmgl_allgather: out[world * count_per_rank] <- every rank's in[count_per_rank], rank order.
 *   mmgl_broadcast: buf[count] of `root` to every rank (DDP's constructor broadcast).  mmgl_comm_destroy: frees the handle.
 *   dtype: MMGL_F32, MMGL_BF16 or MMGL_COMM_I64 (token ids of the eval gather).  Collectives of one communicator must be issued
 *   in the same order on every rank (mmgl_amd.distributed.DataParallelEngine issues its buckets in index order).
 *   RCCL is resolved at run time (the librccl.so the process already holds, else the system's); without it these six return
 *   MMGL_ERR_UNSUPPORTED and every other entry point is unaffected.  mmgl_amd's own trainer reaches the same RCCL through
 *   torch.distributed's "nccl" backend; these are the calls for a host program that has no torch.distributed. */
enum { MMGL_COMM_I64 = 2 };
int mmgl_comm_unique_id(void* out128);
int mmgl_comm_init(int rank, int world, const void* unique_id, void** comm);
int mmgl_allreduce_sum(void* comm, void* buf, size_t count, int dtype, void* stream);
int mmgl_allgather(void* comm, const void* in, void* out, size_t count_per_rank, int dtype, void* stream);
int mmgl_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream);
int mmgl_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* MMGL_HIP_H */
