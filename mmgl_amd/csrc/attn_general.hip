// General (unfused) attention core for gfx950: everything MPTAttention.forward can be asked for that the fused kernels leave out --
//   attention-probability dropout, layer_head_mask, output_attentions --
// replaces: reference model/modelling_cross_attention.py:206-271 (scores, additive finfo.min masks + the max(., finfo.min) clamp :226-235,
//           softmax :229-235, layer_head_mask :237-244, attn_weights_reshaped :246-254, nn.functional.dropout(attn_weights) :256,
//           bmm with V :258) for BOTH call sites of the class: the gated cross-attention layers (key mask only) and the decoder's
//           causal self-attention (causal AND key mask).
// These options are inert on every BASELINE.json config (OPT / Llama attention_dropout = 0, no head masks in run_generation.py), so this
// path is written for exactness and generality, not speed: fp32 arithmetic on the VALU, two passes over the keys (row log-sum-exp
// first, exact probabilities second), any S, D <= 128.  The fused kernels (xattn.hip, selfattn32.hip) keep every other call.
//
//   P   = softmax_s(mask(q . k_s))                      mask: key_valid[b, s] (and s <= t when causal); a query row WITHOUT any
//                                                        allowed key is uniform over all S keys (every score equals finfo.min)
//   W   = head_mask[h] * P                               (what output_attentions returns, :246-254)
//   Wd  = W * keep(b, h, t, s) / (1 - p)                 keep = counter hash of (seed, ((b H + h) T + t) S + s) >= p 2^32: the same
//                                                        hash the LayerNorm / gated-residual dropouts use; regenerated in backward
//   O   = Wd V
// Backward: dWd = dO V^T, delta_t = sum_s dWd Wd, dS = P (head_mask keep / (1 - p) dWd - delta) (x 0.5 on rows without an allowed
// key: autograd's split at the torch.max tie, as in xattn.hip), dQ = dS K, dK = dS^T Q, dV = Wd^T dO.
#include "common.h"
#include <math.h>

namespace {

constexpr int AG_KC = 64;          // keys per chunk (lane = key)
constexpr int AG_ROWS = 64;        // query rows per forward / dQ block (16 per wave)
constexpr int AG_RB = 32;          // query rows per chunk of the dK / dV block

struct AGArgs {
    const void *q, *k, *v, *dout;
    const uint8_t* key_valid;      // [B, S]
    const float* head_mask;        // [H] or NULL
    void *out, *probs;             // out [B, T, H D]; probs [B, H, T, S] or NULL
    float* lse;                    // [B, H, T]
    float* delta;                  // [B, H, T] (backward)
    void *dq, *dk, *dv;
    int B, H, T, S, D, causal;
    float p_drop;
    unsigned long long seed;
};

__device__ __forceinline__ float ag_keep_scale(const AGArgs& a, uint32_t thr, int b, int h, int t, int s) {
    if (a.p_drop <= 0.f) return 1.f;
    const uint64_t idx = (((uint64_t)b * a.H + h) * a.T + t) * (uint64_t)a.S + s;
    return mmgl_hash32(a.seed, idx) < thr ? 0.f : 1.f / (1.f - a.p_drop);
}
__device__ __forceinline__ uint32_t ag_thr(float p) { return (uint32_t)fminf(p * 4294967296.f, 4294967295.f); }

// rows [r0, r0 + nrows) x D of a [.., ld]-pitched tensor -> LDS [nrows][D + 1] fp32 (rows past `limit` as zeros)
template <typename T>
__device__ __forceinline__ void ag_stage(float* dst, const T* src, size_t ld, int r0, int nrows, int limit, int D) {
    for (int i = threadIdx.x; i < nrows * D; i += blockDim.x) {
        const int r = i / D, d = i - r * D;
        dst[r * (D + 1) + d] = (r0 + r < limit) ? (float)src[(size_t)(r0 + r) * ld + d] : 0.f;
    }
}

__device__ __forceinline__ float ag_dot(const float* x, const float* y, int D) {
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(x[d], y[d], s);
    return s;
}

// ---------------------------------------------------------------------------------------------------------------- forward
template <typename T> __global__ __launch_bounds__(256) void ag_fwd_kernel(AGArgs a) {
    extern __shared__ float sm[];
    const int D = a.D, DP = D + 1;
    float* Qs = sm;                         // [AG_ROWS][DP]
    float* Ks = Qs + AG_ROWS * DP;          // [AG_KC][DP]
    float* Vs = Ks + AG_KC * DP;            // [AG_KC][DP]
    const int nrb = (a.T + AG_ROWS - 1) / AG_ROWS;
    const int rb = blockIdx.x % nrb, bh = blockIdx.x / nrb, h = bh % a.H, b = bh / a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ld = (size_t)a.H * D;
    const T* q = (const T*)a.q + (size_t)b * a.T * ld + (size_t)h * D;
    const T* k = (const T*)a.k + (size_t)b * a.S * ld + (size_t)h * D;
    const T* v = (const T*)a.v + (size_t)b * a.S * ld + (size_t)h * D;
    const uint8_t* kv = a.key_valid + (size_t)b * a.S;
    const int r0 = rb * AG_ROWS;
    const float hm = a.head_mask ? a.head_mask[h] : 1.f;
    const uint32_t thr = ag_thr(a.p_drop);
    ag_stage(Qs, q, ld, r0, AG_ROWS, a.T, D);
    const int last_row = min(r0 + AG_ROWS, a.T) - 1;
    const int nch = a.causal ? (last_row / AG_KC + 1) : (a.S + AG_KC - 1) / AG_KC;
    float m[16], l[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { m[i] = -INFINITY; l[i] = 0.f; }
    // pass 1: running max / sum of every row
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        ag_stage(Ks, k, ld, c * AG_KC, AG_KC, a.S, D);
        __syncthreads();
        const int s = c * AG_KC + lane;
        const bool kvalid = s < a.S && kv[s];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tr = wave + 4 * i, t = r0 + tr;
            const bool ok = kvalid && (!a.causal || s <= t) && t < a.T;
            const float sc = ok ? ag_dot(Qs + tr * DP, Ks + lane * DP, D) : -INFINITY;
            const float mn = fmaxf(m[i], wave_max(sc));
            if (mn > -INFINITY) {
                l[i] = l[i] * __expf(m[i] - mn) + wave_sum(ok ? __expf(sc - mn) : 0.f);
                m[i] = mn;
            }
        }
    }
    float lse[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        lse[i] = l[i] > 0.f ? m[i] + __logf(l[i]) : INFINITY;           // +inf: no allowed key -> uniform over the S keys
        const int t = r0 + wave + 4 * i;
        if (lane == 0 && t < a.T) a.lse[((size_t)b * a.H + h) * a.T + t] = lse[i];
    }
    // pass 2: exact probabilities, optional W output, dropout, O = Wd V
    float acc[16][2];
    bool uniform_any = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i][0] = acc[i][1] = 0.f; uniform_any |= lse[i] == INFINITY && r0 + wave + 4 * i < a.T; }
    const float uni = 1.f / (float)a.S;
    // (a row without an allowed key spreads over ALL keys, and W is written for every key: then every chunk is walked)
    const int nch2 = (a.probs || __syncthreads_or(uniform_any ? 1 : 0)) ? (a.S + AG_KC - 1) / AG_KC : nch;
    const int lq = lane < D ? lane : 0, lq1 = 64 + lane < D ? 64 + lane : 0;
    for (int c = 0; c < nch2; ++c) {
        __syncthreads();
        ag_stage(Ks, k, ld, c * AG_KC, AG_KC, a.S, D);
        ag_stage(Vs, v, ld, c * AG_KC, AG_KC, a.S, D);
        __syncthreads();
        const int s = c * AG_KC + lane;
        const bool kvalid = s < a.S && kv[s];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int tr = wave + 4 * i, t = r0 + tr;
            if (t >= a.T) continue;                                     // (wave-uniform)
            const bool ok = kvalid && (!a.causal || s <= t);
            float p;
            if (lse[i] == INFINITY) p = s < a.S ? uni : 0.f;
            else p = ok ? __expf(ag_dot(Qs + tr * DP, Ks + lane * DP, D) - lse[i]) : 0.f;
            const float w = hm * p;
            if (a.probs && s < a.S) ((T*)a.probs)[(((size_t)b * a.H + h) * a.T + t) * a.S + s] = (T)w;
            const float wd = s < a.S ? w * ag_keep_scale(a, thr, b, h, t, s) : 0.f;
            for (int ss = 0; ss < AG_KC; ++ss) {
                const float ws = __shfl(wd, ss);
                acc[i][0] = fmaf(ws, Vs[ss * DP + lq], acc[i][0]);
                if (D > 64) acc[i][1] = fmaf(ws, Vs[ss * DP + lq1], acc[i][1]);
            }
        }
    }
    T* o = (T*)a.out + (size_t)b * a.T * ld + (size_t)h * D;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = r0 + wave + 4 * i;
        if (t >= a.T) continue;
        if (lane < D) o[(size_t)t * ld + lane] = (T)acc[i][0];
        if (64 + lane < D) o[(size_t)t * ld + 64 + lane] = (T)acc[i][1];
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward: dQ (+ delta)
template <typename T> __global__ __launch_bounds__(256) void ag_bwd_dq_kernel(AGArgs a) {
    extern __shared__ float sm[];
    const int D = a.D, DP = D + 1;
    float* Qs = sm;                         // [AG_ROWS][DP]
    float* Gs = Qs + AG_ROWS * DP;          // dO rows
    float* Ks = Gs + AG_ROWS * DP;          // [AG_KC][DP]
    float* Vs = Ks + AG_KC * DP;
    const int nrb = (a.T + AG_ROWS - 1) / AG_ROWS;
    const int rb = blockIdx.x % nrb, bh = blockIdx.x / nrb, h = bh % a.H, b = bh / a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ld = (size_t)a.H * D;
    const T* q = (const T*)a.q + (size_t)b * a.T * ld + (size_t)h * D;
    const T* g = (const T*)a.dout + (size_t)b * a.T * ld + (size_t)h * D;
    const T* k = (const T*)a.k + (size_t)b * a.S * ld + (size_t)h * D;
    const T* v = (const T*)a.v + (size_t)b * a.S * ld + (size_t)h * D;
    const uint8_t* kv = a.key_valid + (size_t)b * a.S;
    const int r0 = rb * AG_ROWS;
    const float hm = a.head_mask ? a.head_mask[h] : 1.f;
    const uint32_t thr = ag_thr(a.p_drop);
    ag_stage(Qs, q, ld, r0, AG_ROWS, a.T, D);
    ag_stage(Gs, g, ld, r0, AG_ROWS, a.T, D);
    const int last_row = min(r0 + AG_ROWS, a.T) - 1;
    const float uni = 1.f / (float)a.S;
    float lse[16], delta[16], acc[16][2];
    bool uniform_any = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = r0 + wave + 4 * i;
        lse[i] = t < a.T ? a.lse[((size_t)b * a.H + h) * a.T + t] : 0.f;
        uniform_any |= lse[i] == INFINITY && t < a.T;
        delta[i] = 0.f;
        acc[i][0] = acc[i][1] = 0.f;
    }
    // a row without an allowed key spreads over ALL keys: such a block walks every chunk even when causal
    const int any_uniform_block = __syncthreads_or(uniform_any ? 1 : 0);
    const int nch = (a.causal && !any_uniform_block) ? (last_row / AG_KC + 1) : (a.S + AG_KC - 1) / AG_KC;
    const int lq = lane < D ? lane : 0, lq1 = 64 + lane < D ? 64 + lane : 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int c = 0; c < nch; ++c) {
            __syncthreads();
            ag_stage(Ks, k, ld, c * AG_KC, AG_KC, a.S, D);
            ag_stage(Vs, v, ld, c * AG_KC, AG_KC, a.S, D);
            __syncthreads();
            const int s = c * AG_KC + lane;
            const bool kvalid = s < a.S && kv[s];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int tr = wave + 4 * i, t = r0 + tr;
                if (t >= a.T) continue;
                const bool ok = kvalid && (!a.causal || s <= t);
                const bool uniform = lse[i] == INFINITY;
                float p;
                if (uniform) p = s < a.S ? uni : 0.f;
                else p = ok ? __expf(ag_dot(Qs + tr * DP, Ks + lane * DP, D) - lse[i]) : 0.f;
                const float ks = s < a.S ? hm * ag_keep_scale(a, thr, b, h, t, s) : 0.f;
                const float dwd = p != 0.f ? ag_dot(Gs + tr * DP, Vs + lane * DP, D) : 0.f;
                if (pass == 0) {
                    delta[i] += wave_sum(dwd * p * ks);
                } else {
                    // a row without any allowed key: every score IS finfo.min after the clamp and torch.max splits the gradient at the
                    // tie -- except on keys masked TWICE (future and padded: finfo.min + finfo.min = -inf, the clamp constant wins outright)
                    const float tie = (a.causal && s > t && !kvalid) ? 0.f : 0.5f;
                    const float ds = (uniform ? tie : 1.f) * p * (ks * dwd - delta[i]);
                    for (int ss = 0; ss < AG_KC; ++ss) {
                        const float dss = __shfl(ds, ss);
                        acc[i][0] = fmaf(dss, Ks[ss * DP + lq], acc[i][0]);
                        if (D > 64) acc[i][1] = fmaf(dss, Ks[ss * DP + lq1], acc[i][1]);
                    }
                }
            }
        }
    }
    T* dq = (T*)a.dq + (size_t)b * a.T * ld + (size_t)h * D;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = r0 + wave + 4 * i;
        if (t >= a.T) continue;
        if (lane == 0) a.delta[((size_t)b * a.H + h) * a.T + t] = delta[i];
        if (lane < D) dq[(size_t)t * ld + lane] = (T)acc[i][0];
        if (64 + lane < D) dq[(size_t)t * ld + 64 + lane] = (T)acc[i][1];
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward: dK, dV
// One block per (b, h, 64-key chunk): loops over the query rows in chunks of AG_RB; per chunk the waves compute Wd and dS of their
// rows (lane = key) into LDS, then every wave folds the chunk into the accumulators of ITS 16 keys (lane = feature).
template <typename T> __global__ __launch_bounds__(256) void ag_bwd_dkv_kernel(AGArgs a) {
    extern __shared__ float sm[];
    const int D = a.D, DP = D + 1;
    float* Ks = sm;                         // [AG_KC][DP]
    float* Vs = Ks + AG_KC * DP;
    float* Qs = Vs + AG_KC * DP;            // [AG_RB][DP]
    float* Gs = Qs + AG_RB * DP;
    float* Ws = Gs + AG_RB * DP;            // [AG_RB][AG_KC]  Wd
    float* Ss = Ws + AG_RB * AG_KC;         // [AG_RB][AG_KC]  dS
    const int nkc = (a.S + AG_KC - 1) / AG_KC;
    const int kc = blockIdx.x % nkc, bh = blockIdx.x / nkc, h = bh % a.H, b = bh / a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ld = (size_t)a.H * D;
    const T* q = (const T*)a.q + (size_t)b * a.T * ld + (size_t)h * D;
    const T* g = (const T*)a.dout + (size_t)b * a.T * ld + (size_t)h * D;
    const T* k = (const T*)a.k + (size_t)b * a.S * ld + (size_t)h * D;
    const T* v = (const T*)a.v + (size_t)b * a.S * ld + (size_t)h * D;
    const uint8_t* kv = a.key_valid + (size_t)b * a.S;
    const float hm = a.head_mask ? a.head_mask[h] : 1.f;
    const uint32_t thr = ag_thr(a.p_drop);
    const float uni = 1.f / (float)a.S;
    ag_stage(Ks, k, ld, kc * AG_KC, AG_KC, a.S, D);
    ag_stage(Vs, v, ld, kc * AG_KC, AG_KC, a.S, D);
    const int s = kc * AG_KC + lane;
    const bool kvalid = s < a.S && kv[s];
    float dk[16][2], dv[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[i][0] = dk[i][1] = dv[i][0] = dv[i][1] = 0.f;
    // (causal: rows before the chunk's first key see none of its keys -- except rows without any allowed key, which are uniform
    // over every key; those exist only when the caller breaks the key-0-valid precondition, so every row chunk is walked)
    for (int r0 = 0; r0 < a.T; r0 += AG_RB) {
        __syncthreads();
        ag_stage(Qs, q, ld, r0, AG_RB, a.T, D);
        ag_stage(Gs, g, ld, r0, AG_RB, a.T, D);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AG_RB / 4; ++i) {
            const int tr = wave + 4 * i, t = r0 + tr;
            float wd = 0.f, ds = 0.f;
            if (t < a.T) {
                const float lse = a.lse[((size_t)b * a.H + h) * a.T + t], delta = a.delta[((size_t)b * a.H + h) * a.T + t];
                const bool ok = kvalid && (!a.causal || s <= t);
                const bool uniform = lse == INFINITY;
                float p;
                if (uniform) p = s < a.S ? uni : 0.f;
                else p = ok ? __expf(ag_dot(Qs + tr * DP, Ks + lane * DP, D) - lse) : 0.f;
                if (p != 0.f) {
                    const float ks = hm * ag_keep_scale(a, thr, b, h, t, s);
                    const float dwd = ag_dot(Gs + tr * DP, Vs + lane * DP, D);
                    wd = p * ks;
                    const float tie = (a.causal && s > t && !kvalid) ? 0.f : 0.5f;      // doubly-masked keys: see the dQ kernel
                    ds = (uniform ? tie : 1.f) * p * (ks * dwd - delta);
                }
            }
            Ws[tr * AG_KC + lane] = wd;
            Ss[tr * AG_KC + lane] = ds;
        }
        __syncthreads();
        for (int tr = 0; tr < AG_RB; ++tr) {
            const int lq = lane < D ? lane : 0, lq1 = 64 + lane < D ? 64 + lane : 0;
            const float g0 = Gs[tr * DP + lq], q0 = Qs[tr * DP + lq];
            const float g1 = D > 64 ? Gs[tr * DP + lq1] : 0.f, q1 = D > 64 ? Qs[tr * DP + lq1] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float wd = Ws[tr * AG_KC + 16 * wave + i], ds = Ss[tr * AG_KC + 16 * wave + i];
                dv[i][0] = fmaf(wd, g0, dv[i][0]);
                dk[i][0] = fmaf(ds, q0, dk[i][0]);
                dv[i][1] = fmaf(wd, g1, dv[i][1]);
                dk[i][1] = fmaf(ds, q1, dk[i][1]);
            }
        }
    }
    T* dkp = (T*)a.dk + (size_t)b * a.S * ld + (size_t)h * D;
    T* dvp = (T*)a.dv + (size_t)b * a.S * ld + (size_t)h * D;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int sk = kc * AG_KC + 16 * wave + i;
        if (sk >= a.S) continue;
        if (lane < D) { dkp[(size_t)sk * ld + lane] = (T)dk[i][0]; dvp[(size_t)sk * ld + lane] = (T)dv[i][0]; }
        if (64 + lane < D) { dkp[(size_t)sk * ld + 64 + lane] = (T)dk[i][1]; dvp[(size_t)sk * ld + 64 + lane] = (T)dv[i][1]; }
    }
}

__global__ __launch_bounds__(256) void ag_mask_kernel(uint8_t* mask, size_t n, float p, unsigned long long seed) {
    const uint32_t thr = ag_thr(p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        mask[i] = (p > 0.f && mmgl_hash32(seed, i) < thr) ? 0 : 1;
}

template <typename K> int ag_lds(K kern, size_t bytes) {
    if (bytes > 160 * 1024) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "attn_general: %zu B of LDS", bytes);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return MMGL_OK;
}

int ag_check(const char* who, int B, int H, int T, int S, int D, int causal, float p, int dtype) {
    MMGL_CHECK_ARG(B > 0 && H > 0 && T > 0 && S > 0 && D > 0, "%s: bad sizes B=%d H=%d T=%d S=%d D=%d", who, B, H, T, S, D);
    if (D > 128) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: head_dim %d > 128", who, D);
    MMGL_CHECK_ARG(!causal || S == T, "%s: causal attention needs S == T (got %d, %d)", who, S, T);
    MMGL_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability %g outside [0, 1)", who, (double)p);
    MMGL_CHECK_ARG(dtype == MMGL_F32 || dtype == MMGL_BF16, "%s: dtype %d", who, dtype);
    return MMGL_OK;
}

}  // namespace

extern "C" int mmgl_attn_general_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, const float* head_mask, void* out,
                                     void* probs, float* lse, int B, int H, int T, int S, int D, int causal, float p_drop,
                                     uint64_t seed, int dtype, void* stream) {
    if (int rc = ag_check("mmgl_attn_general_fwd", B, H, T, S, D, causal, p_drop, dtype)) return rc;
    MMGL_CHECK_ARG(q && k && v && key_valid && out && lse, "mmgl_attn_general_fwd: null pointer");
    AGArgs a{};
    a.q = q; a.k = k; a.v = v; a.key_valid = key_valid; a.head_mask = head_mask; a.out = out; a.probs = probs; a.lse = lse;
    a.B = B; a.H = H; a.T = T; a.S = S; a.D = D; a.causal = causal; a.p_drop = p_drop; a.seed = seed;
    const size_t lds = (size_t)(AG_ROWS + 2 * AG_KC) * (D + 1) * sizeof(float);
    const dim3 grid(B * H * cdiv(T, AG_ROWS));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) {
        if (int rc = ag_lds(ag_fwd_kernel<bf16>, lds)) return rc;
        hipLaunchKernelGGL(ag_fwd_kernel<bf16>, grid, dim3(256), lds, st, a);
    } else {
        if (int rc = ag_lds(ag_fwd_kernel<float>, lds)) return rc;
        hipLaunchKernelGGL(ag_fwd_kernel<float>, grid, dim3(256), lds, st, a);
    }
    MMGL_CHECK_LAUNCH("attn_general_fwd");
    return MMGL_OK;
}

extern "C" size_t mmgl_attn_general_bwd_workspace(int B, int H, int T) { return align_up((size_t)B * H * T * sizeof(float), 256); }

extern "C" int mmgl_attn_general_bwd(const void* dout, const void* q, const void* k, const void* v, const float* lse, const uint8_t* key_valid,
                                     const float* head_mask, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes, int B,
                                     int H, int T, int S, int D, int causal, float p_drop, uint64_t seed, int dtype, void* stream) {
    if (int rc = ag_check("mmgl_attn_general_bwd", B, H, T, S, D, causal, p_drop, dtype)) return rc;
    MMGL_CHECK_ARG(dout && q && k && v && lse && key_valid && dq && dk && dv && workspace, "mmgl_attn_general_bwd: null pointer");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_attn_general_bwd_workspace(B, H, T), "mmgl_attn_general_bwd: workspace too small");
    AGArgs a{};
    a.q = q; a.k = k; a.v = v; a.dout = dout; a.key_valid = key_valid; a.head_mask = head_mask; a.lse = (float*)lse; a.delta = (float*)workspace;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.B = B; a.H = H; a.T = T; a.S = S; a.D = D; a.causal = causal; a.p_drop = p_drop; a.seed = seed;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds1 = (size_t)(2 * AG_ROWS + 2 * AG_KC) * (D + 1) * sizeof(float);
    const size_t lds2 = ((size_t)(2 * AG_KC + 2 * AG_RB) * (D + 1) + 2 * AG_RB * AG_KC) * sizeof(float);
    const dim3 g1(B * H * cdiv(T, AG_ROWS)), g2(B * H * cdiv(S, AG_KC));
    if (dtype == MMGL_BF16) {
        if (int rc = ag_lds(ag_bwd_dq_kernel<bf16>, lds1)) return rc;
        if (int rc = ag_lds(ag_bwd_dkv_kernel<bf16>, lds2)) return rc;
        hipLaunchKernelGGL(ag_bwd_dq_kernel<bf16>, g1, dim3(256), lds1, st, a);
        hipLaunchKernelGGL(ag_bwd_dkv_kernel<bf16>, g2, dim3(256), lds2, st, a);
    } else {
        if (int rc = ag_lds(ag_bwd_dq_kernel<float>, lds1)) return rc;
        if (int rc = ag_lds(ag_bwd_dkv_kernel<float>, lds2)) return rc;
        hipLaunchKernelGGL(ag_bwd_dq_kernel<float>, g1, dim3(256), lds1, st, a);
        hipLaunchKernelGGL(ag_bwd_dkv_kernel<float>, g2, dim3(256), lds2, st, a);
    }
    MMGL_CHECK_LAUNCH("attn_general_bwd");
    return MMGL_OK;
}

// the keep mask of mmgl_attn_general_fwd / _bwd as bytes [B, H, T, S] (1 = kept): a debug / test entry point -- the parity tests feed it
// to the oracle so that both sides drop the same probabilities
extern "C" int mmgl_attn_dropout_mask(uint8_t* mask, int B, int H, int T, int S, float p_drop, uint64_t seed, void* stream) {
    MMGL_CHECK_ARG(mask && B > 0 && H > 0 && T > 0 && S > 0 && p_drop >= 0.f && p_drop < 1.f, "mmgl_attn_dropout_mask: bad arguments");
    const size_t n = (size_t)B * H * T * S;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ag_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, n, p_drop, seed);
    MMGL_CHECK_LAUNCH("attn_dropout_mask");
    return MMGL_OK;
}
