// Masked neighbor cross-attention core for gfx950:  O = softmax(max(QK^T + M, finfo.min)) V
//
// Replaces the ATen op chain of MPTAttention.forward (reference model/modelling_cross_attention.py:206-271)
// plus the materialised [B,1,T,S] additive mask of _expand_mask (:68-79).
//
// Shape regime: S = n_neighbors * n_tokens is SHORT (16..128 keys), T is long (640..2176 query
// rows), D in {64,128}.  The whole K/V of one (batch, head) fits in LDS, so there is no online-softmax loop:
// one pass, the full score row of a query lives in registers.  The kernel is HBM-bound (AI ~ 58..121 FLOP/B
// vs a ridge of ~312): what matters is streaming Q in and O out in 16-byte lane accesses with no materialised
// head transpose; MFMA only does the dense QK^T / PV contractions.
//
// Lane geometry (wave64, v_mfma 16x16): lane = (x = l & 15, g = l >> 4).
//   "swapped" products  S^T = K Q^T  and  O^T = V^T P^T  make every lane own ONE query row (x) and, of that
//   row, the keys / channels 4g..4g+3 of every 16-block.  Row max / sum = in-lane reduce + two xor-shuffles
//   (16, 32).  The accumulator layout of S^T is *already* a legal B-operand layout for the second product
//   (any k-permutation is legal if both operands share it), so P never moves between lanes: the V^T (or K^T)
//   operand image in LDS is simply stored in the matching key order.
//
// LDS images (filled once per workgroup, read by all 4 waves):
//   row image  R[sb][dc][lane][8] : lane (x,g) -> X[s = 16 sb + x][d = 32 dc + 8 g + 0..7]      (A operand, K or V)
//   t   image  Tm[db][ks][lane][8]: lane (x,g), e -> X[s = 16(2ks + (e>>2)) + 4g + (e&3)][d = 16 db + x]
// Both are "fragment linear": a fragment read is one conflict-free ds_read_b128 per lane.
// fp32 activations skip the t image (LDS budget) and gather it from the row image with scalar reads.
#include "attn_common.h"
#include <atomic>
#include <mutex>

namespace {

// store one query row's NDB blocks held as 4 channels per lane per block
template <typename T, typename C>
__device__ __forceinline__ void store_row_tiles(__amdgpu_buffer_rsrc_t ro, int t, int T_, uint32_t row_bytes, int g,
                                                const typename Elem<T>::v4 (&o)[C::NDB]) {
    if constexpr (sizeof(T) == 2 && C::NDB % 2 == 0) {
        const uint32_t rb = (t < T_) ? (uint32_t)t * row_bytes : 0x80000000u;   // past any slab, and no wrap-around when the column offset is added
#pragma unroll
        for (int db = 0; db < C::NDB; db += 2) store_pair_bf16(ro, rb, db, g, o[db], o[db + 1]);
    } else {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) buf_store_v4<T>(ro, row_off<T, C>(t, row_bytes, db * 16 + g * 4), o[db]);
    }
}

// ============================================================================================ forward
// LDS: K as a fragment-linear row image; V as a row-major padded image read through ds_read_b64_tr_b16 (bf16) or as a
// second row image gathered with scalar reads (fp32).  A workgroup covers `rows_per_wg` query rows of one (b, h); its 4
// waves take 16*QT-row tiles round-robin, and each wave issues the NEXT tile's Q loads before it computes the current
// one, so every wave keeps HBM requests in flight across its MFMA / softmax / store phases.
#ifndef MMGL_XATTN_MINWAVES
#define MMGL_XATTN_MINWAVES 1
#endif
template <typename T, int D, int NSB>
__global__ __launch_bounds__(256, MMGL_XATTN_MINWAVES) void xattn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, const uint8_t* __restrict__ valid,
                                                        T* __restrict__ out, float* __restrict__ lse, int B, int H,
                                                        int T_, int S, int rows_per_wg, int nchunk) {
    typedef XC<T, D, NSB> C;
    typedef typename Elem<T>::v8 v8;
    constexpr int TILE = 16 * C::QT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Kf = (T*)smem;
    T* Vi = Kf + C::ROWIMG;                                   // row-major image (bf16) or row image (f32)
    uint8_t* vld = (uint8_t*)(Vi + (C::TIMG ? C::RMIMG : C::ROWIMG));

    const int vid = xcd_remap(blockIdx.x, B * H * nchunk);
    const int bh = vid / nchunk, chunk = vid % nchunk;
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;

    const int wg_begin = chunk * rows_per_wg;
    const int wg_end = min(wg_begin + rows_per_wg, T_);
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t slab = (uint32_t)(((size_t)(T_ - 1) * HD + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse + (size_t)bh * T_, (uint32_t)(T_ * sizeof(float)));

    // first tile's Q goes out before the K/V staging so its latency overlaps the LDS fill
    int t0 = wg_begin + wave * TILE;
    v8 qn[C::QT][C::NDC];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc)
            qn[qt][dc] = buf_load8<T>(rq, row_off<T, C>((t0 < wg_end) ? t0 + qt * 16 + x : T_, row_bytes, dc * 32 + g * 8));

    {   // K, V and the key mask in ONE memory round trip (ImageStage); the mask byte is only loaded here, its select sits at the store
        const uint32_t slab_kv = (uint32_t)(((size_t)(S - 1) * HD + D) * sizeof(T));
        const __amdgpu_buffer_rsrc_t rk = make_rsrc(k + (size_t)b * S * HD + h * D, slab_kv);
        const __amdgpu_buffer_rsrc_t rv = make_rsrc(v + (size_t)b * S * HD + h * D, slab_kv);
        ImageStage<T, C, 256> ks_, vs_;
        const int vi = min((int)threadIdx.x, C::SPAD - 1);
        const uint8_t vraw = valid[(size_t)b * S + min(vi, S - 1)];
        ks_.load(rk, row_bytes);
        vs_.load(rv, row_bytes);
        ks_.store_row(Kf);
        if constexpr (C::TIMG) vs_.store_rowmajor(Vi);
        else vs_.store_row(Vi);
        if ((int)threadIdx.x < C::SPAD) vld[threadIdx.x] = (vi < S) ? vraw : (uint8_t)0;
    }
    __syncthreads();

    uint32_t vlo, vhi, elo, ehi;
    lane_key_bits<C>(vld, g, S, vlo, vhi, elo, ehi);
    const bool any_valid = __ballot((vlo | vhi) != 0) != 0ull;
    f32x4 bias[C::NSB];
    key_bias<C>(vlo, vhi, bias);

    // Software pipeline (per wave, in VMEM issue order):  stores(i-1), loads(i+1), compute(i).
    // vmcnt is an in-order counter shared by loads and stores, so the wait for tile i's loads also waits for every
    // older store; issuing a tile's stores one compute phase late (right before the next loads) means both have had a
    // whole MFMA/softmax phase to complete and the wait is free, instead of a store round trip per iteration.
    typedef typename Elem<T>::v4 v4;
    v4 ost[C::QT][C::NDB];
    float lst[C::QT];
    int tprev = T_;                       // row index past the slab: the first round's stores are dropped by the bounds check
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        lst[qt] = 0.f;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) ost[qt][db] = vzero<v4>();
    }
    for (; t0 < wg_end; t0 += 4 * TILE) {
        if constexpr (!C::HOIST) asm volatile("" ::: "memory");
        v8 qf[C::QT][C::NDC];
        const int tn = t0 + 4 * TILE;
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) qf[qt][dc] = qn[qt][dc];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            const int t = tprev + qt * 16 + x;
            store_row_tiles<T, C>(ro, t, T_, row_bytes, g, ost[qt]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, lst[qt]), rl, (g == 0) ? (uint32_t)t * 4u : OOB, 0, 0);
        }
        tprev = t0;
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc)
                qn[qt][dc] = buf_load8<T>(rq, row_off<T, C>((tn < wg_end) ? tn + qt * 16 + x : T_, row_bytes, dc * 32 + g * 8));

        f32x4 sacc[C::QT][C::NSB];
#pragma unroll
        for (int sb = 0; sb < C::NSB; ++sb) {
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) sacc[qt][sb] = bias[sb];          // masked keys start (and stay) at -inf
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                const v8 kf = *(const v8*)(Kf + rf_idx<C>(sb, dc, lane));
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(sacc[qt][sb], kf, qf[qt][dc]);
            }
        }

        v8 pf[C::QT][C::NKS];
        float linv[C::QT], lsev[C::QT];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            float m, l = 0.f;
            if (any_valid) {
                m = fmaxf(fmaxf(sacc[qt][0][0], sacc[qt][0][1]), fmaxf(sacc[qt][0][2], sacc[qt][0][3]));
#pragma unroll
                for (int sb = 1; sb < C::NSB; ++sb)
                    m = fmaxf(m, fmaxf(fmaxf(sacc[qt][sb][0], sacc[qt][sb][1]), fmaxf(sacc[qt][sb][2], sacc[qt][sb][3])));
                m = xg_max(m);
                const float m2 = m * LOG2E;
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[qt][sb][r], LOG2E, -m2));
                        sacc[qt][sb][r] = p;
                        l += p;
                    }
            } else {   // every key masked: all scores clamp to finfo.min -> uniform over the S real keys
                m = 0.f;
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = bit64(elo, ehi, sb * 4 + r) ? 1.f : 0.f;
                        sacc[qt][sb][r] = p;
                        l += p;
                    }
            }
            l = xg_sum(l);
            linv[qt] = __builtin_amdgcn_rcpf(l);
            lsev[qt] = m + __logf(l);
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) pf[qt][ks] = pack8<T>(sacc[qt][2 * ks], sacc[qt][2 * ks + 1]);
        }

        f32x4 oacc[C::QT][C::NDB];
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) oacc[qt][db] = vzero<f32x4>();
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                v8 vt;
                if constexpr (C::TIMG) vt = rm_tfrag_tr16<C>(Vi, db, ks, lane);
                else vt = load_tfrag<T, C>(Vi, Vi, db, ks, lane);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(oacc[qt][db], vt, pf[qt][ks]);
            }
        }

#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            lst[qt] = lsev[qt];
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) ost[qt][db] = cvt4<T>(oacc[qt][db] * linv[qt]);
        }
    }
    // drain: the last tile's outputs
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int t = tprev + qt * 16 + x;
        store_row_tiles<T, C>(ro, t, T_, row_bytes, g, ost[qt]);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, lst[qt]), rl, (g == 0) ? (uint32_t)t * 4u : OOB, 0, 0);
    }
}

// ============================================================================================ backward 1: dQ (+ row dots)
// Same streaming structure as forward.  P is recomputed from the saved LSE; delta_t = sum_s P dP (== dO.O, so O is
// never re-read); dS = P (dP - delta); dQ^T = K^T dS^T.  delta is written for the dK/dV kernel.
// LDS (bf16): K row-major padded (row fragments for S^T AND tr16 fragments for K^T), V fragment-linear row image.
// LDS (fp32): K, V row images; K^T gathered with scalar reads.
template <typename T, int D, int NSB>
__global__ __launch_bounds__(256) void xattn_bwd_dq_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                           const T* __restrict__ k, const T* __restrict__ v,
                                                           const float* __restrict__ lse, const uint8_t* __restrict__ valid,
                                                           T* __restrict__ dq, float* __restrict__ delta, int B, int H,
                                                           int T_, int S, int rows_per_wg, int nchunk) {
    typedef XC<T, D, NSB, 2> C;
    typedef typename Elem<T>::v8 v8;
    constexpr int TILE = 16 * C::QT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ki = (T*)smem;                                          // row-major (bf16) / row image (f32)
    T* Vf = Ki + (C::TIMG ? C::RMIMG : C::ROWIMG);
    uint8_t* vld = (uint8_t*)(Vf + C::ROWIMG);

    const int vid = xcd_remap(blockIdx.x, B * H * nchunk);
    const int bh = vid / nchunk, chunk = vid % nchunk;
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;

    const int wg_begin = chunk * rows_per_wg;
    const int wg_end = min(wg_begin + rows_per_wg, T_);
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t slab = (uint32_t)(((size_t)(T_ - 1) * HD + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(dout + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dq + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse + (size_t)bh * T_, (uint32_t)(T_ * sizeof(float)));
    const __amdgpu_buffer_rsrc_t rdl = make_rsrc(delta + (size_t)bh * T_, (uint32_t)(T_ * sizeof(float)));

    // same memory pipeline as the forward kernel: bounds-checked buffer accesses (no branches), per wave in VMEM order
    //   stores(i-1), loads(i+1), compute(i)
    int t0 = wg_begin + wave * TILE;
    v8 qn[C::QT][C::NDC], gn[C::QT][C::NDC];
    float lsn[C::QT];
    auto request = [&](int tbase, bool live) __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            const int t = live ? tbase + qt * 16 + x : T_;
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                qn[qt][dc] = buf_load8<T>(rq, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
                gn[qt][dc] = buf_load8<T>(rg, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
            }
            lsn[qt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (uint32_t)t * 4u, 0, 0));
        }
    };
    request(t0, t0 < wg_end);

    if constexpr (C::TIMG) stage_rowmajor_image<T, C>(Ki, k + (size_t)b * S * HD + h * D, HD, S);
    else stage_row_image<T, C>(Ki, k + (size_t)b * S * HD + h * D, HD, S);
    stage_row_image<T, C>(Vf, v + (size_t)b * S * HD + h * D, HD, S);
    for (int i = threadIdx.x; i < C::SPAD; i += blockDim.x) vld[i] = (i < S) ? valid[(size_t)b * S + i] : 0;
    __syncthreads();

    uint32_t vlo, vhi, elo, ehi;
    lane_key_bits<C>(vld, g, S, vlo, vhi, elo, ehi);
    const bool any_valid = __ballot((vlo | vhi) != 0) != 0ull;
    const float uni = 1.f / (float)S;
    const float tie = any_valid ? 1.f : 0.5f;                 // autograd's 50/50 split at the torch.max tie
    f32x4 bias[C::NSB];
    key_bias<C>(vlo, vhi, bias);                              // 0 / -inf per key: the mask enters as the MFMA C input

    typedef typename Elem<T>::v4 v4;
    v4 ost[C::QT][C::NDB];
    float dst[C::QT];
    int tprev = T_;
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        dst[qt] = 0.f;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) ost[qt][db] = vzero<v4>();
    }
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            const int t = tprev + qt * 16 + x;
            store_row_tiles<T, C>(rd, t, T_, row_bytes, g, ost[qt]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, dst[qt]), rdl, (g == 0) ? (uint32_t)t * 4u : OOB, 0, 0);
        }
    };
    for (; t0 < wg_end; t0 += 4 * TILE) {
        if constexpr (!C::HOIST) asm volatile("" ::: "memory");
        v8 qf[C::QT][C::NDC], gf[C::QT][C::NDC];
        float l2[C::QT];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            l2[qt] = lsn[qt] * LOG2E;
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) { qf[qt][dc] = qn[qt][dc]; gf[qt][dc] = gn[qt][dc]; }
        }
        flush();
        tprev = t0;
        request(t0 + 4 * TILE, t0 + 4 * TILE < wg_end);

        v8 dsf[C::QT][C::NKS];
        float dlt[C::QT];
        {
            f32x4 sacc[C::QT][C::NSB], pacc[C::QT][C::NSB];
#pragma unroll
            for (int sb = 0; sb < C::NSB; ++sb) {
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) { sacc[qt][sb] = bias[sb]; pacc[qt][sb] = vzero<f32x4>(); }
#pragma unroll
                for (int dc = 0; dc < C::NDC; ++dc) {
                    v8 kf;
                    if constexpr (C::TIMG) kf = rm_rowfrag<T, C>(Ki, sb, dc, lane);
                    else kf = *(const v8*)(Ki + rf_idx<C>(sb, dc, lane));
                    const v8 vf = *(const v8*)(Vf + rf_idx<C>(sb, dc, lane));
#pragma unroll
                    for (int qt = 0; qt < C::QT; ++qt) {
                        mma16(sacc[qt][sb], kf, qf[qt][dc]);
                        mma16(pacc[qt][sb], vf, gf[qt][dc]);
                    }
                }
            }
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) {
                float dl = 0.f;
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p;
                        if (any_valid) p = __builtin_amdgcn_exp2f(fmaf(sacc[qt][sb][r], LOG2E, -l2[qt]));    // masked key: exp2(-inf) = 0
                        else p = bit64(elo, ehi, sb * 4 + r) ? uni : 0.f;
                        sacc[qt][sb][r] = p;
                        dl += p * pacc[qt][sb][r];
                    }
                dl = xg_sum(dl);
                dlt[qt] = dl;
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[qt][sb][r] = tie * sacc[qt][sb][r] * (pacc[qt][sb][r] - dl);
#pragma unroll
                for (int ks = 0; ks < C::NKS; ++ks) dsf[qt][ks] = pack8<T>(sacc[qt][2 * ks], sacc[qt][2 * ks + 1]);
            }
        }
        f32x4 acc[C::QT][C::NDB];
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) acc[qt][db] = vzero<f32x4>();
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                v8 kt;
                if constexpr (C::TIMG) kt = rm_tfrag_tr16<C>(Ki, db, ks, lane);
                else kt = load_tfrag<T, C>(Ki, Ki, db, ks, lane);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(acc[qt][db], kt, dsf[qt][ks]);
            }
        }
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            dst[qt] = dlt[qt];
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) ost[qt][db] = cvt4<T>(acc[qt][db]);
        }
    }
    flush();                                                  // the last tile's outputs
}

// ============================================================================================ backward 2: dK, dV partials
// One wave per (batch, head, 32-key group, T-chunk).  Non-swapped products put lane = key column, so the
// probabilities come out of the S = Q K^T accumulator directly in B-operand layout for the contraction over t:
//     dV^T[d][s] += dO^T[d][t] P[t][s]      dK^T[d][s] += Q^T[d][t] dS[t][s]
// No row reductions are needed here: LSE comes from forward, delta from the dQ kernel.  The transposed Q / dO
// operands are read back from a wave-private row-major LDS tile.
template <typename T, int D>
__global__ __launch_bounds__(64) void xattn_bwd_dkv_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                           const T* __restrict__ k, const T* __restrict__ v,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           const uint8_t* __restrict__ valid, float* __restrict__ dk_part,
                                                           float* __restrict__ dv_part, int B, int H, int T_, int S,
                                                           int nsg, int rows_per_chunk, int nchunk) {
    typedef XC<T, D, 2> C;                       // one 32-key group = 2 key blocks
    typedef typename Elem<T>::v8 v8;
    constexpr int DLD = C::DPAD + (sizeof(T) == 2 ? 8 : 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Kb = (T*)smem;
    T* Vb = Kb + C::ROWIMG;
    T* Qt = Vb + C::ROWIMG;
    T* Gt = Qt + 32 * DLD;

    const int vid = xcd_remap(blockIdx.x, B * H * nsg * nchunk);
    const int chunk = vid % nchunk;
    const int sg = (vid / nchunk) % nsg;
    const int bh = vid / (nchunk * nsg);
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int s0 = sg * 32;

    stage_row_image<T, C>(Kb, k + ((size_t)b * S + s0) * HD + h * D, HD, S - s0);
    stage_row_image<T, C>(Vb, v + ((size_t)b * S + s0) * HD + h * D, HD, S - s0);

    const int lane = threadIdx.x, x = lane & 15, g = lane >> 4;
    bool any = false;
    for (int s = lane; s < S; s += 64) any |= valid[(size_t)b * S + s] != 0;
    const bool any_valid = __ballot(any) != 0ull;
    const float uni = 1.f / (float)S, tie = any_valid ? 1.f : 0.5f;
    bool vs[2], es[2];
#pragma unroll
    for (int sbl = 0; sbl < 2; ++sbl) {
        const int s = s0 + sbl * 16 + x;
        es[sbl] = s < S;
        vs[sbl] = es[sbl] && valid[(size_t)b * S + s] != 0;
    }
    __syncthreads();

    f32x4 dva[C::NDB][2], dka[C::NDB][2];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) { dva[db][sbl] = vzero<f32x4>(); dka[db][sbl] = vzero<f32x4>(); }

    const T* qb = q + (size_t)b * T_ * HD + h * D;
    const T* gb = dout + (size_t)b * T_ * HD + h * D;
    const float* lb = lse + (size_t)bh * T_;
    const float* dlb = delta + (size_t)bh * T_;
    const int row_begin = chunk * rows_per_chunk, row_end = min(row_begin + rows_per_chunk, T_);

    for (int t0 = row_begin; t0 < row_end; t0 += 32) {
        v8 qa[2][C::NDC], ga[2][C::NDC];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                const int t = t0 + tb * 16 + x;
                qa[tb][dc] = load_qfrag<T, C>(qb, t, T_, HD, dc * 32 + g * 8);
                ga[tb][dc] = load_qfrag<T, C>(gb, t, T_, HD, dc * 32 + g * 8);
                *(v8*)(Qt + (tb * 16 + x) * DLD + dc * 32 + g * 8) = qa[tb][dc];
                *(v8*)(Gt + (tb * 16 + x) * DLD + dc * 32 + g * 8) = ga[tb][dc];
            }
        f32x4 pr[2][2], dsr[2][2];               // [tb][sbl]
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            float lt[4], dt[4];
            bool tv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + tb * 16 + g * 4 + r;
                tv[r] = t < row_end;
                lt[r] = tv[r] ? lb[t] : 0.f;
                dt[r] = tv[r] ? dlb[t] : 0.f;
            }
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                f32x4 sa = vzero<f32x4>(), pa = vzero<f32x4>();
#pragma unroll
                for (int dc = 0; dc < C::NDC; ++dc) {
                    const v8 kb = *(const v8*)(Kb + rf_idx<C>(sbl, dc, lane));
                    const v8 vb = *(const v8*)(Vb + rf_idx<C>(sbl, dc, lane));
                    mma16(sa, qa[tb][dc], kb);
                    mma16(pa, ga[tb][dc], vb);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p;
                    if (any_valid) p = (tv[r] && vs[sbl]) ? __expf(sa[r] - lt[r]) : 0.f;
                    else p = (tv[r] && es[sbl]) ? uni : 0.f;
                    pr[tb][sbl][r] = p;
                    dsr[tb][sbl][r] = tie * p * (pa[r] - dt[r]);
                }
            }
        }
        v8 pB[2], dsB[2];
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) {
            pB[sbl] = pack8<T>(pr[0][sbl], pr[1][sbl]);
            dsB[sbl] = pack8<T>(dsr[0][sbl], dsr[1][sbl]);
        }
        __syncthreads();
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
            v8 gT, qT;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = (e >> 2) * 16 + g * 4 + (e & 3);
                gT[e] = Gt[row * DLD + db * 16 + x];
                qT[e] = Qt[row * DLD + db * 16 + x];
            }
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                mma16(dva[db][sbl], gT, pB[sbl]);
                mma16(dka[db][sbl], qT, dsB[sbl]);
            }
        }
        __syncthreads();
    }
    // partials [nchunk][B][S][H*D] fp32; lane (x = key, g) owns channels 16 db + 4 g .. +3
#pragma unroll
    for (int sbl = 0; sbl < 2; ++sbl) {
        const int s = s0 + sbl * 16 + x;
        if (s < S) {
            const size_t off = (((size_t)chunk * B + b) * S + s) * HD + h * D + g * 4;
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) {
                *(f32x4*)(dk_part + off + db * 16) = dka[db][sbl];
                *(f32x4*)(dv_part + off + db * 16) = dva[db][sbl];
            }
        }
    }
}

// ============================================================================================ backward, fused (bf16, D <= 64, S <= 64)
// dQ, dK and dV out of ONE pass over Q and dO: a wave owns (batch, head, T-chunk) and ALL keys, streams 32-row tiles
// (requested one tile ahead, bounds by buffer descriptors) and per tile
//   1. swapped products S^T = K Q^T, dP^T = V dO^T (lane = query row): P from the saved LSE, delta = sum_s P dP in-lane,
//      dS = P (dP - delta), dQ^T = K^T dS^T with dS^T straight out of the accumulators (as xattn_bwd_dq_kernel);
//   2. P and dS (bf16) dropped row-major into a wave-private LDS tile next to the tile's Q and dO rows, and read back with
//      ds_read_b64_tr_b16 as the operands of the contraction over t:  dV^T[d][s] += dO^T[d][t] P[t][s],
//      dK^T[d][s] += Q^T[d][t] dS[t][s]  (accumulators live in registers for the whole chunk).
// HBM traffic = the algorithmic minimum: Q and dO read once, dQ written once, no delta array, no second pass over Q / dO
// (the two-kernel path reads them twice), and with one chunk per (batch, head) dK / dV are written once in bf16 with no
// fp32 partials.  One wave per workgroup: no barrier anywhere, every dK / dV element of a chunk is produced by one wave.
template <int D, int NSB>
__global__ __launch_bounds__(64) void xattn_bwd_fused_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ q,
                                                             const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                             const float* __restrict__ lse, const uint8_t* __restrict__ valid,
                                                             bf16* __restrict__ dq, bf16* __restrict__ dk, bf16* __restrict__ dv,
                                                             float* __restrict__ dk_part, float* __restrict__ dv_part, int B, int H,
                                                             int T_, int S, int rows_per_chunk, int nchunk) {
    typedef bf16 T;
    typedef XC<T, D, NSB, 2> C;
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static_assert(C::QT == 2, "fused backward: a tile is one 32-row contraction step");
    constexpr int LDT = C::DPAD + 16;              // Q / dO tile row stride (elements)
    constexpr int LDP = C::SPAD + 16;              // P / dS tile row stride
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ki = (T*)smem;                              // row-major padded: row fragments (S^T) and tr16 fragments (K^T)
    T* Vf = Ki + C::RMIMG;                         // fragment-linear row image
    T* Qt = Vf + C::ROWIMG;                        // [32][LDT]
    T* Gt = Qt + 32 * LDT;
    T* Pt = Gt + 32 * LDT;                         // [32][LDP]
    T* DSt = Pt + 32 * LDP;
    uint8_t* vld = (uint8_t*)(DSt + 32 * LDP);

    const int lane = threadIdx.x, x = lane & 15, g = lane >> 4;
    const int vid = xcd_remap(blockIdx.x, B * H * nchunk);
    const int bh = vid / nchunk, chunk = vid % nchunk;
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int row_begin = chunk * rows_per_chunk, row_end = min(row_begin + rows_per_chunk, T_);

    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t slab = (uint32_t)(((size_t)(T_ - 1) * HD + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(dout + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dq + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse + (size_t)bh * T_, (uint32_t)(T_ * sizeof(float)));

    // two tiles in flight: a register set is re-requested (two tiles ahead) as soon as its swapped products are issued
    v8 qA[2][C::NDC], gA[2][C::NDC], qB[2][C::NDC], gB[2][C::NDC];
    float lA[2], lB[2];
    auto request = [&](int tbase, v8 (&qn)[2][C::NDC], v8 (&gn)[2][C::NDC], float (&lsn)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int t = (tbase < row_end) ? tbase + qt * 16 + x : T_;      // past the chunk: every access falls outside the slab
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                qn[qt][dc] = buf_load8<T>(rq, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
                gn[qt][dc] = buf_load8<T>(rg, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
            }
            lsn[qt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (uint32_t)t * 4u, 0, 0));
        }
    };
    request(row_begin, qA, gA, lA);
    request(row_begin + 32, qB, gB, lB);

    {   // K, V and the key mask in ONE memory round trip (ImageStage: a one-wave workgroup has nothing else to hide 17 of them)
        static_assert(C::SPAD <= 64, "fused backward: one mask byte per lane");
        const uint32_t slab_kv = (uint32_t)(((size_t)(S - 1) * HD + D) * sizeof(T));
        const __amdgpu_buffer_rsrc_t rk = make_rsrc(k + (size_t)b * S * HD + h * D, slab_kv);
        const __amdgpu_buffer_rsrc_t rv = make_rsrc(v + (size_t)b * S * HD + h * D, slab_kv);
        ImageStage<T, C, 64> ks_, vs_;
        const uint8_t vraw = valid[(size_t)b * S + min(lane, S - 1)];
        ks_.load(rk, row_bytes);
        vs_.load(rv, row_bytes);
        ks_.store_rowmajor(Ki);
        vs_.store_row(Vf);
        if (lane < C::SPAD) vld[lane] = (lane < S) ? vraw : (uint8_t)0;
    }
    __syncthreads();

    uint32_t vlo, vhi, elo, ehi;
    lane_key_bits<C>(vld, g, S, vlo, vhi, elo, ehi);
    const bool any_valid = __ballot((vlo | vhi) != 0) != 0ull;
    const float uni = 1.f / (float)S;
    const float tie = any_valid ? 1.f : 0.5f;                 // autograd's 50/50 split at the torch.max tie
    f32x4 bias[C::NSB];
    key_bias<C>(vlo, vhi, bias);

    f32x4 dva[C::NDB][C::NSB], dka[C::NDB][C::NSB];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int sb = 0; sb < C::NSB; ++sb) { dva[db][sb] = vzero<f32x4>(); dka[db][sb] = vzero<f32x4>(); }

    v4 ost[2][C::NDB];
    int tprev = T_;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) ost[qt][db] = vzero<v4>();
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) store_row_tiles<T, C>(rd, tprev + qt * 16 + x, T_, row_bytes, g, ost[qt]);
    };

    auto step = [&](int t0, v8 (&qf)[2][C::NDC], v8 (&gf)[2][C::NDC], float (&lsn)[2]) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");                        // the K / V fragments are re-read from LDS every tile (no hoisting: registers)
        float l2[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) l2[qt] = (t0 + qt * 16 + x < row_end) ? lsn[qt] * LOG2E : INFINITY;     // row past the chunk: p = 0
        flush();                                              // VMEM order per wave: stores(i-1), compute(i) ... loads(i+2)
        tprev = t0;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                *(v8*)(Qt + (qt * 16 + x) * LDT + dc * 32 + g * 8) = qf[qt][dc];
                *(v8*)(Gt + (qt * 16 + x) * LDT + dc * 32 + g * 8) = gf[qt][dc];
            }

        v8 dsf[2][C::NKS];
        {
            f32x4 sacc[2][C::NSB], pacc[2][C::NSB];
#pragma unroll
            for (int sb = 0; sb < C::NSB; ++sb) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) { sacc[qt][sb] = bias[sb]; pacc[qt][sb] = vzero<f32x4>(); }
#pragma unroll
                for (int dc = 0; dc < C::NDC; ++dc) {
                    const v8 kf = rm_rowfrag<T, C>(Ki, sb, dc, lane);
                    const v8 vf = *(const v8*)(Vf + rf_idx<C>(sb, dc, lane));
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        mma16(sacc[qt][sb], kf, qf[qt][dc]);
                        mma16(pacc[qt][sb], vf, gf[qt][dc]);
                    }
                }
            }
            request(t0 + 64, qf, gf, lsn);                    // this set's rows are consumed (LDS tile + MFMA operands): next-but-one tile
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const bool live = t0 + qt * 16 + x < row_end;
                float dl = 0.f;
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float p;
                        if (any_valid) p = __builtin_amdgcn_exp2f(fmaf(sacc[qt][sb][r], LOG2E, -l2[qt]));    // masked key: exp2(-inf) = 0
                        else p = (live && bit64(elo, ehi, sb * 4 + r)) ? uni : 0.f;
                        sacc[qt][sb][r] = p;
                        dl += p * pacc[qt][sb][r];
                    }
                    *(v4*)(Pt + (qt * 16 + x) * LDP + sb * 16 + g * 4) = cvt4<T>(sacc[qt][sb]);
                }
                dl = xg_sum(dl);
                v4 dsb[C::NSB];
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb) {
                    f32x4 d4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) d4[r] = tie * sacc[qt][sb][r] * (pacc[qt][sb][r] - dl);
                    dsb[sb] = cvt4<T>(d4);
                    *(v4*)(DSt + (qt * 16 + x) * LDP + sb * 16 + g * 4) = dsb[sb];
                }
#pragma unroll
                for (int ks = 0; ks < C::NKS; ++ks) {
                    const v4 lo = dsb[2 * ks], hi = dsb[2 * ks + 1];
                    dsf[qt][ks] = v8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            }
        }
        {
            f32x4 acc[2][C::NDB];
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) {
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) acc[qt][db] = vzero<f32x4>();
#pragma unroll
                for (int ks = 0; ks < C::NKS; ++ks) {
                    const v8 kt = rm_tfrag_tr16<C>(Ki, db, ks, lane);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) mma16(acc[qt][db], kt, dsf[qt][ks]);
                }
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) ost[qt][db] = cvt4<T>(acc[qt][db]);
        }
        // contraction over the tile's 32 rows; LDS ops of one wave execute in order, the fences only stop compiler reordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
            const int trow = 4 * g + (x >> 2), tcol = (x & 3) * 4;
            v8 pB[C::NSB], dsB[C::NSB];
#pragma unroll
            for (int sb = 0; sb < C::NSB; ++sb) {
                const bf16* pp = Pt + trow * LDP + sb * 16 + tcol;
                const bf16* pd = DSt + trow * LDP + sb * 16 + tcol;
                const bf16x4 p0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pp);
                const bf16x4 p1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pp + 16 * LDP));
                const bf16x4 d0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pd);
                const bf16x4 d1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pd + 16 * LDP));
                pB[sb] = v8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                dsB[sb] = v8{d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
            }
#pragma unroll
            for (int db = 0; db < C::NDB; ++db) {
                const bf16* pg = Gt + trow * LDT + db * 16 + tcol;
                const bf16* pq = Qt + trow * LDT + db * 16 + tcol;
                const bf16x4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pg);
                const bf16x4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pg + 16 * LDT));
                const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pq);
                const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pq + 16 * LDT));
                const v8 gT = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                const v8 qT = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
                for (int sb = 0; sb < C::NSB; ++sb) {
                    mma16(dva[db][sb], gT, pB[sb]);
                    mma16(dka[db][sb], qT, dsB[sb]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int t0 = row_begin; t0 < row_end; t0 += 64) {
        step(t0, qA, gA, lA);
        if (t0 + 32 < row_end) step(t0 + 32, qB, gB, lB);
    }
    flush();                                                  // the last tile's dQ

#pragma unroll
    for (int sb = 0; sb < C::NSB; ++sb) {
        const int s = sb * 16 + x;
        if (s < S) {
            if (nchunk == 1) {
                const size_t off = ((size_t)b * S + s) * HD + h * D + g * 4;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    store4<T>(dk + off + db * 16, dka[db][sb]);
                    store4<T>(dv + off + db * 16, dva[db][sb]);
                }
            } else {
                const size_t off = (((size_t)chunk * B + b) * S + s) * HD + h * D + g * 4;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    *(f32x4*)(dk_part + off + db * 16) = dka[db][sb];
                    *(f32x4*)(dv_part + off + db * 16) = dva[db][sb];
                }
            }
        }
    }
}

// ============================================================================================ backward, fused, keys split over waves (bf16)
// The one-pass backward above for the shapes whose dK / dV accumulators do not fit one wave: head_dim 128 and / or 64 < S <= 128
// (config 5: D = 128, S = 128 -- 512 accumulator registers per wave in the one-wave form).  A workgroup = NW = NSB / 2 waves owns
// (batch, head, T-chunk); wave w owns KEYS 32 w .. 32 w + 31 for S^T, dP^T, P, dS, dK, dV (128 accumulator registers at D = 128)
// and CHANNELS of dQ: the contraction dQ^T = K^T dS^T runs over all keys, so the waves trade dS^T (bf16, already in B-operand
// layout: 2 KiB per wave and tile through LDS) and each finishes NDB / NW channel blocks -- no fp32 partials, no atomics.
// Per 32-row tile:  the waves load disjoint (row half, channel chunk) pieces of Q / dO (HBM and L2 see every byte once) into a
// shared, parity-double-buffered LDS tile [B1]; swapped products for the wave's keys (lane = query row), P from the saved LSE,
// partial delta = sum over the wave's keys of P dP to LDS [B2]; delta = sum of the NW partials, dS = P (dP - delta), dS^T
// fragments to LDS [B3]; dQ for the wave's channels; P / dS through a wave-private tile and ds_read_b64_tr_b16 into
// dV^T += dO^T P, dK^T += Q^T dS.  Three barriers per tile; a wave that runs ahead can only touch the other parity's Q / dO tile.
// HBM traffic = the algorithmic minimum (Q, dO read once, dQ written once, K / V / dK / dV once per chunk).
// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. the Q / dO rows requested two
// tiles ahead and the dQ stores of the previous tile, three times per tile
#define XW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int D, int NSB>
__global__ __launch_bounds__(32 * NSB) void xattn_bwd_fusedw_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ q,
                                                                    const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                                    const float* __restrict__ lse, const uint8_t* __restrict__ valid,
                                                                    bf16* __restrict__ dq, bf16* __restrict__ dk, bf16* __restrict__ dv,
                                                                    float* __restrict__ dk_part, float* __restrict__ dv_part, int B,
                                                                    int H, int T_, int S, int rows_per_chunk, int nchunk) {
    typedef bf16 T;
    typedef XC<T, D, NSB, 2> C;
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    constexpr int NW = NSB / 2, NT = 64 * NW;      // waves, threads
    constexpr int DBW = C::NDB / NW;               // 16-channel blocks of dQ per wave
    constexpr int IPW = 2 * C::NDC / NW;           // (row half, 32-channel chunk) load items per wave
    static_assert(NSB % 2 == 0 && C::NDB % NW == 0 && (2 * C::NDC) % NW == 0 && D % 32 == 0, "fusedw: shape");
    constexpr int LDT = C::DPAD + 16;              // Q / dO tile row stride (elements)
    constexpr int LDP = 32 + 16;                   // wave-private P / dS tile row stride: 32 keys
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ki = (T*)smem;                              // row-major padded K: read ONCE, by ds_read_b64_tr_b16, for the K^T fragments below
    T* QG = Ki + C::RMIMG;                         // [parity][Q | dO][32][LDT]
    T* PD = QG + 2 * 2 * 32 * LDT;                 // [wave][P | dS][32][LDP]
    T* DSX = PD + NW * 2 * 32 * LDP;               // [ks = wave][qt][lane][8]: dS^T B-operand fragments of every wave
    float* dpart = (float*)(DSX + NW * 2 * 64 * 8);   // [wave][32] partial deltas
    float* LSEt = dpart + NW * 32;                 // [parity][32] the tile's saved log-sum-exp
    uint8_t* vld = (uint8_t*)(LSEt + 2 * 32);

    const int lane = threadIdx.x & 63, x = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int vid = xcd_remap(blockIdx.x, B * H * nchunk);
    const int bh = vid / nchunk, chunk = vid % nchunk;
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int row_begin = chunk * rows_per_chunk, row_end = min(row_begin + rows_per_chunk, T_);

    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T));
    const uint32_t slab = (uint32_t)(((size_t)(T_ - 1) * HD + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(dout + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dq + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse + (size_t)bh * T_, (uint32_t)(T_ * sizeof(float)));

    // this wave's pieces of a tile: item i = IPW wave + j -> row half qt = i & 1, channel chunk dc = i >> 1.  Two register sets:
    // the tile after next is requested as soon as a set has been dropped into the shared LDS tile
    // (the tile's 32 lse values ride along as ONE dword per lane and go through LDS like the rows: kept in registers across the loop
    // and scaled at the top of a step, hipcc rotated them with v_mov behind s_waitcnt vmcnt(0) at the back edge -- the prefetch drained)
    v8 qA[IPW], gA[IPW], qB[IPW], gB[IPW];
    uint32_t lA, lB;
    auto request = [&](int tbase, v8 (&qn)[IPW], v8 (&gn)[IPW], uint32_t& lsn) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = IPW * wave + j, qt = i & 1, dc = i >> 1;
            const int t = (tbase < row_end) ? tbase + qt * 16 + x : T_;      // past the chunk: every access falls outside the slab
            qn[j] = buf_load8<T>(rq, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
            gn[j] = buf_load8<T>(rg, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
        }
        lsn = __builtin_amdgcn_raw_buffer_load_b32(rl, (tbase < row_end && lane < 32) ? (uint32_t)(tbase + lane) * 4u : OOB, 0, 0);
    };
    request(row_begin, qA, gA, lA);
    request(row_begin + 32, qB, gB, lB);

    // K and V of this wave's keys live in REGISTERS for the whole chunk (one wave per SIMD: the register file is there, the LDS port
    // is the scarce resource -- re-reading them every tile was 96 of the 312 KiB a tile moved through LDS):
    //   kf / vf [sbl][dc]  row fragments (A operands of S^T = K Q^T, dP^T = V dO^T), straight from global memory in fragment layout;
    //   kt [dbl][ks]       K^T fragments of this wave's dQ channels over ALL keys, by transposed reads of a row-major K image
    v8 kf[2][C::NDC], vf[2][C::NDC], kt[DBW][C::NKS];
    {
        const uint32_t slab_kv = (uint32_t)(((size_t)(S - 1) * HD + D) * sizeof(T));
        const __amdgpu_buffer_rsrc_t rk = make_rsrc(k + (size_t)b * S * HD + h * D, slab_kv);
        const __amdgpu_buffer_rsrc_t rv = make_rsrc(v + (size_t)b * S * HD + h * D, slab_kv);
        ImageStage<T, C, NT> ks_;
        const int tid = (int)threadIdx.x;
        const uint8_t vraw = valid[(size_t)b * S + min(tid, S - 1)];
        ks_.load(rk, row_bytes);
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl)
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {            // rows past S fall outside the descriptor: zero fragments
                const uint32_t off = row_off<T, C>((2 * wave + sbl) * 16 + x, row_bytes, dc * 32 + g * 8);
                kf[sbl][dc] = buf_load8<T>(rk, off);
                vf[sbl][dc] = buf_load8<T>(rv, off);
            }
        ks_.store_rowmajor(Ki);
        if (tid < C::SPAD) vld[tid] = (tid < S) ? vraw : (uint8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int dbl = 0; dbl < DBW; ++dbl)
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) kt[dbl][ks] = rm_tfrag_tr16<C>(Ki, wave * DBW + dbl, ks, lane);
    // (a load issued before a loop and first used inside it stays pending in hipcc's scoreboard at the loop header: every trip would
    // wait for it with counts that also drain the prefetch -- have the fragments read once here)
#pragma unroll
    for (int sbl = 0; sbl < 2; ++sbl)
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) asm volatile("" ::"v"(kf[sbl][dc]), "v"(vf[sbl][dc]));
#pragma unroll
    for (int dbl = 0; dbl < DBW; ++dbl)
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) asm volatile("" ::"v"(kt[dbl][ks]));

    uint32_t vlo, vhi, elo, ehi;
    lane_key_bits<C>(vld, g, S, vlo, vhi, elo, ehi);
    const bool any_valid = __ballot((vlo | vhi) != 0) != 0ull;
    const float uni = 1.f / (float)S;
    const float tie = any_valid ? 1.f : 0.5f;                 // autograd's 50/50 split at the torch.max tie
    f32x4 bias[2];                                            // this wave's two 16-key blocks
#pragma unroll
    for (int sbl = 0; sbl < 2; ++sbl)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[sbl][r] = bit64(vlo, vhi, (2 * wave + sbl) * 4 + r) ? 0.f : -INFINITY;

    f32x4 dva[C::NDB][2], dka[C::NDB][2];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) { dva[db][sbl] = vzero<f32x4>(); dka[db][sbl] = vzero<f32x4>(); }

    v4 ost[2][DBW];
    int tprev = T_;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dbl = 0; dbl < DBW; ++dbl) ost[qt][dbl] = vzero<v4>();
    auto flush = [&]() __attribute__((always_inline)) {       // dQ of the previous tile: this wave's channel blocks
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int t = tprev + qt * 16 + x;
            const uint32_t rb = (t < row_end) ? (uint32_t)t * row_bytes : 0x80000000u;      // (a tile past the chunk computes zeros: never stored)
            if constexpr (DBW % 2 == 0) {
#pragma unroll
                for (int dbl = 0; dbl < DBW; dbl += 2) store_pair_bf16(rd, rb, wave * DBW + dbl, g, ost[qt][dbl], ost[qt][dbl + 1]);
            } else {
#pragma unroll
                for (int dbl = 0; dbl < DBW; ++dbl)
                    buf_store_v4<T>(rd, rb + (uint32_t)(((wave * DBW + dbl) * 16 + g * 4) * sizeof(T)), ost[qt][dbl]);
            }
        }
    };

    // dV^T += dO^T P, dK^T += Q^T dS over the 32 rows of the tile of parity `par` for this wave's keys: P / dS from the wave-private
    // tiles (LDS ops of a wave run in order), dO^T / Q^T from the shared tile, all by ds_read_b64_tr_b16.  (Run one tile LATE, inside the
    // next tile's first segment, with parity-double-buffered P / dS tiles: measured 162.9 against 158.4 us -- no gain, not kept.)
    auto contract = [&](int par) __attribute__((always_inline)) {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const T* Qt = QG + par * 2 * 32 * LDT;
        const T* Gt = Qt + 32 * LDT;
        const T* Pt = PD + wave * 2 * 32 * LDP;
        const T* DSt = Pt + 32 * LDP;
        const int trow = 4 * g + (x >> 2), tcol = (x & 3) * 4;
        v8 pB[2], dsB[2];
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) {
            const bf16* pp = Pt + trow * LDP + sbl * 16 + tcol;
            const bf16* pd = DSt + trow * LDP + sbl * 16 + tcol;
            const bf16x4 p0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pp);
            const bf16x4 p1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pp + 16 * LDP));
            const bf16x4 d0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pd);
            const bf16x4 d1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pd + 16 * LDP));
            pB[sbl] = v8{p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
            dsB[sbl] = v8{d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
            const bf16* pg = Gt + trow * LDT + db * 16 + tcol;
            const bf16* pq = Qt + trow * LDT + db * 16 + tcol;
            const bf16x4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pg);
            const bf16x4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pg + 16 * LDT));
            const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pq);
            const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pq + 16 * LDT));
            const v8 gT = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
            const v8 qT = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                mma16(dva[db][sbl], gT, pB[sbl]);
                mma16(dka[db][sbl], qT, dsB[sbl]);
            }
        }
    };

    auto step = [&](int t0, int par, v8 (&qn)[IPW], v8 (&gn)[IPW], uint32_t& lsn) __attribute__((always_inline)) {
        T* Qt = QG + par * 2 * 32 * LDT;
        T* Gt = Qt + 32 * LDT;
        T* Pt = PD + wave * 2 * 32 * LDP;
        T* DSt = Pt + 32 * LDP;
        flush();                                              // VMEM order per wave: stores(i-1), compute(i) ... loads(i+2)
        tprev = t0;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int i = IPW * wave + j, qt = i & 1, dc = i >> 1;
            *(v8*)(Qt + (qt * 16 + x) * LDT + dc * 32 + g * 8) = qn[j];
            *(v8*)(Gt + (qt * 16 + x) * LDT + dc * 32 + g * 8) = gn[j];
        }
        if (wave == 0 && lane < 32) ((uint32_t*)LSEt)[par * 32 + lane] = lsn;
        XW_BARRIER();                                         // [B1] the tile's Q / dO rows are in LDS
        request(t0 + 64, qn, gn, lsn);                        // this set's registers are free: the tile after next
        float l2[2];                                          // rows past T: Q = dO = 0 and lse reads 0 -> finite p, dP = 0, dS = 0
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) l2[qt] = LSEt[par * 32 + qt * 16 + x] * LOG2E;

        f32x4 sacc[2][2], pacc[2][2];                         // [qt][sbl]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) { sacc[qt][sbl] = bias[sbl]; pacc[qt][sbl] = vzero<f32x4>(); }
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) {
            v8 qf[2], gf[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                qf[qt] = *(const v8*)(Qt + (qt * 16 + x) * LDT + dc * 32 + g * 8);
                gf[qt] = *(const v8*)(Gt + (qt * 16 + x) * LDT + dc * 32 + g * 8);
            }
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    mma16(sacc[qt][sbl], kf[sbl][dc], qf[qt]);
                    mma16(pacc[qt][sbl], vf[sbl][dc], gf[qt]);
                }
        }
        // P for this wave's keys, partial delta
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const bool live = t0 + qt * 16 + x < row_end;
            float dl = 0.f;
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv;
                    if (any_valid) pv = __builtin_amdgcn_exp2f(fmaf(sacc[qt][sbl][r], LOG2E, -l2[qt]));    // masked key: exp2(-inf) = 0
                    else pv = (live && bit64(elo, ehi, (2 * wave + sbl) * 4 + r)) ? uni : 0.f;
                    sacc[qt][sbl][r] = pv;
                    dl += pv * pacc[qt][sbl][r];
                }
                *(v4*)(Pt + (qt * 16 + x) * LDP + sbl * 16 + g * 4) = cvt4<T>(sacc[qt][sbl]);
            }
            dl = xg_sum(dl);
            if (g == 0) dpart[wave * 32 + qt * 16 + x] = dl;
        }
        XW_BARRIER();                                         // [B2] every wave's partial delta
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float dl = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) dl += dpart[w * 32 + qt * 16 + x];
            v4 dsb[2];
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                f32x4 d4;
#pragma unroll
                for (int r = 0; r < 4; ++r) d4[r] = tie * sacc[qt][sbl][r] * (pacc[qt][sbl][r] - dl);
                dsb[sbl] = cvt4<T>(d4);
                *(v4*)(DSt + (qt * 16 + x) * LDP + sbl * 16 + g * 4) = dsb[sbl];
            }
            const v8 f = {dsb[0][0], dsb[0][1], dsb[0][2], dsb[0][3], dsb[1][0], dsb[1][1], dsb[1][2], dsb[1][3]};
            *(v8*)(DSX + ((wave * 2 + qt) * 64 + lane) * 8) = f;     // B-operand fragment of key step ks = wave
        }
        XW_BARRIER();                                         // [B3] dS^T of all keys
        {
            f32x4 acc[2][DBW];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int dbl = 0; dbl < DBW; ++dbl) acc[qt][dbl] = vzero<f32x4>();
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                v8 dsf[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) dsf[qt] = *(const v8*)(DSX + ((ks * 2 + qt) * 64 + lane) * 8);
#pragma unroll
                for (int dbl = 0; dbl < DBW; ++dbl)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) mma16(acc[qt][dbl], kt[dbl][ks], dsf[qt]);
            }
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int dbl = 0; dbl < DBW; ++dbl) ost[qt][dbl] = cvt4<T>(acc[qt][dbl]);
        }
        contract(par);
    };
    // Both register sets every trip (row_end is uniform over the workgroup: every wave takes every barrier).  A chunk of an odd number
    // of tiles runs one empty tile (loads fall outside the descriptor, p = 0, nothing stored): with the second step conditional, hipcc
    // resolves the loop-carried lse registers with a v_mov behind s_waitcnt vmcnt(0) -- the whole prefetch drained every trip.
    for (int t0 = row_begin; t0 < row_end; t0 += 64) {
        step(t0, 0, qA, gA, lA);
        step(t0 + 32, 1, qB, gB, lB);
    }
    flush();                                                  // the last tile's dQ

#pragma unroll
    for (int sbl = 0; sbl < 2; ++sbl) {
        const int s_ = (2 * wave + sbl) * 16 + x;
        if (s_ < S) {
            if (nchunk == 1) {
                const size_t off = ((size_t)b * S + s_) * HD + h * D + g * 4;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    store4<T>(dk + off + db * 16, dka[db][sbl]);
                    store4<T>(dv + off + db * 16, dva[db][sbl]);
                }
            } else {
                const size_t off = (((size_t)chunk * B + b) * S + s_) * HD + h * D + g * 4;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    *(f32x4*)(dk_part + off + db * 16) = dka[db][sbl];
                    *(f32x4*)(dv_part + off + db * 16) = dva[db][sbl];
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, T* __restrict__ outp,
                                                              size_t n4, size_t chunk_stride4, int nchunk) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 a = ((const f32x4*)part)[i];
        for (int c = 1; c < nchunk; ++c) a += ((const f32x4*)part)[i + c * chunk_stride4];
        store4<T>(outp + i * 4, a);
    }
}

// ============================================================================================ host dispatch
struct BwdPlan { int nsg, nchunk, rows_per_chunk; size_t delta_off, dk_off, dv_off, total; };

// (the two-kernel backward below the fused ones: 32 keys per dK/dV wave.  Round 5 removed its bf16-only 64-key variant,
// xattn_bwd_dkv64_kernel: since the one-pass kernels took every BASELINE shape it only served bf16 with S > 128 or head_dim 16 / 32
// with S > 64, which the generic kernel covers)
inline constexpr bool use_fused_bwd() { return true; }

// the multi-wave one-pass backward (xattn_bwd_fusedw_kernel): bf16, head_dim 64 / 128, up to 128 keys, and not the shapes the one-wave
// fused kernel takes (D <= 64 with S <= 64)
inline bool use_fusedw(int S, int D, size_t esz) { return esz == 2 && S <= 128 && (D == 128 || (D == 64 && S > 64)); }

BwdPlan bwd_plan(int B, int H, int T, int S, int D, size_t esz = 2) {
    BwdPlan p;
    const bool fw = use_fusedw(S, D, esz);
    p.nsg = fw ? 1 : (S + 31) / 32;
    long units = (long)B * H * p.nsg;
    // fusedw: one workgroup per CU (LDS); a chunk costs a set of fp32 dK / dV partials, so chunks only while (batch, head) pairs
    // alone do not fill the chip
    int nchunk = fw ? (units >= 256 ? 1 : (int)((512 + units - 1) / units)) : (int)((2048 + units - 1) / units);
    int maxchunk = (T + 63) / 64;
    if (nchunk > maxchunk) nchunk = maxchunk;
    if (nchunk < 1) nchunk = 1;
    p.rows_per_chunk = ((T + nchunk - 1) / nchunk + 31) / 32 * 32;
    p.nchunk = (T + p.rows_per_chunk - 1) / p.rows_per_chunk;
    size_t hd = (size_t)H * D;
    p.delta_off = 0;
    p.dk_off = align_up((size_t)B * H * T * sizeof(float), 256);
    p.dv_off = p.dk_off + align_up((size_t)p.nchunk * B * S * hd * sizeof(float), 256);
    p.total = p.dv_off + align_up((size_t)p.nchunk * B * S * hd * sizeof(float), 256);
    return p;
}

int tune_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// rows_per_wg / nchunk: about `target` workgroups in flight (all resident in one round: 2-5 per CU), each workgroup a
// whole number of 4-wave tile rounds where possible -- and never ONE workgroup per (batch, head) when the rows allow two: with
// B H >= 1024 equal workgroups the launch is whole rounds of identical workgroups that stage K / V at the same moment and drain
// their last stores at the same moment (4 per CU at D = 64: 2048 pairs = two rounds).  Two workgroups per pair (K / V staged twice,
// the second time out of L2): 92.1 -> 85.5 us at B = 64 on one box, 87.4 -> 82.8 on another (round 6,
// profiles/r6_xattn_fwd_chunks_sweep.txt).  Measured and NOT adopted: three / four / six per pair (83.5 / 88 / 94), two of unequal
// length (55 / 60 / 70 % of the rows in the first: 87.5 / 90.8 / 99.5 -- the long ones set the tail), 16-row tiles per wave with
// five or six workgroups per CU (88-89).
void fwd_geometry(int B, int H, int T, int tile_rows, int& rows_per_wg, int& nchunk) {
    static const int target = tune_env("MMGL_XATTN_TARGET_WGS", 512);
    static const int min_chunks = tune_env("MMGL_XATTN_MIN_CHUNKS", 2);
    const int bh = B * H;
    int nc = (target + bh / 2) / bh;
    const int maxc = (T + tile_rows - 1) / tile_rows;
    if (nc < min_chunks && T >= 8 * tile_rows) nc = min_chunks;
    if (nc > maxc) nc = maxc;
    if (nc < 1) nc = 1;
    int rows = ((T + nc - 1) / nc + tile_rows - 1) / tile_rows * tile_rows;
    rows_per_wg = rows;
    nchunk = (T + rows - 1) / rows;
}

// hipFuncSetAttribute ONCE per (kernel, device), to the hardware maximum (the attribute is a launch limit, not an allocation: the
// launch's own dynamic size decides occupancy): at the reference's batch the step is launch-bound and this call sat in front of every
// cross-attention launch above 48 KiB.  Lookup is lock-free; the first use of a (kernel, device) pair takes a mutex, so two threads
// with different LDS sizes cannot leave the attribute below what the table says (it is never lowered: there is one value).
template <typename K> int set_lds(K kern, size_t bytes) {
    constexpr size_t kMaxLds = 160 * 1024;
    if (bytes > kMaxLds) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "xattn: S*D needs %zu B of LDS (> 160 KiB)", bytes);
    if (bytes <= 48 * 1024) return MMGL_OK;
    struct Slot { std::atomic<const void*> fn{nullptr}; int dev = -1; };
    static Slot slots[256];
    static std::atomic<int> used{0};
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void* fn = (const void*)kern;
    const int n = used.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (slots[i].fn.load(std::memory_order_relaxed) == fn && slots[i].dev == dev) return MMGL_OK;
    std::lock_guard<std::mutex> lock(mu);
    const int n2 = used.load(std::memory_order_acquire);
    for (int i = n; i < n2; ++i)
        if (slots[i].fn.load(std::memory_order_relaxed) == fn && slots[i].dev == dev) return MMGL_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
    if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (n2 < 256) {                                     // a full table only costs the call again next time
        slots[n2].dev = dev;
        slots[n2].fn.store(fn, std::memory_order_relaxed);
        used.store(n2 + 1, std::memory_order_release);
    }
    return MMGL_OK;
}

template <typename T, int D, int NSB>
int launch_fwd(const void* q, const void* k, const void* v, const uint8_t* valid, void* out, float* lse, int B, int H,
               int T_, int S, hipStream_t st) {
    typedef XC<T, D, NSB> C;
    int rpw, nchunk;
    fwd_geometry(B, H, T_, 16 * C::QT, rpw, nchunk);
    size_t lds = sizeof(T) * (C::ROWIMG + (C::TIMG ? C::RMIMG : C::ROWIMG)) + C::SPAD;
    auto kern = xattn_fwd_kernel<T, D, NSB>;
    int rc = set_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(B * H * nchunk), dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, valid,
                       (T*)out, lse, B, H, T_, S, rpw, nchunk);
    MMGL_CHECK_LAUNCH("xattn_fwd");
    return MMGL_OK;
}

template <typename T, int D, int NSB>
int launch_bwd(const void* dout, const void* q, const void* k, const void* v, const float* lse, const uint8_t* valid,
               void* dq, void* dk, void* dv, char* ws, int B, int H, int T_, int S, hipStream_t st) {
    typedef XC<T, D, NSB, 2> C;
    const BwdPlan p = bwd_plan(B, H, T_, S, D, sizeof(T));
    float* delta = (float*)(ws + p.delta_off);
    float* dkp = (float*)(ws + p.dk_off);
    float* dvp = (float*)(ws + p.dv_off);
    if constexpr (sizeof(T) == 2 && (D == 128 || D == 64) && NSB <= 8 && !(D <= 64 && NSB <= 4)) {
        // one pass over Q / dO with the keys split over NSB / 2 waves (p.nsg == 1: a workgroup holds all keys)
        constexpr int NW = NSB / 2, LDT = C::DPAD + 16;
        const size_t lds = sizeof(bf16) * (C::RMIMG + 2 * 2 * 32 * LDT + NW * 2 * 32 * 48 + NW * 2 * 64 * 8) + (NW * 32 + 2 * 32) * sizeof(float) + C::SPAD;
        auto kern = xattn_bwd_fusedw_kernel<D, NSB>;
        int rc = set_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(B * H * p.nchunk), dim3(64 * NW), lds, st, (const bf16*)dout, (const bf16*)q, (const bf16*)k,
                           (const bf16*)v, lse, valid, (bf16*)dq, (bf16*)dk, (bf16*)dv, dkp, dvp, B, H, T_, S, p.rows_per_chunk, p.nchunk);
        MMGL_CHECK_LAUNCH("xattn_bwd_fusedw");
        if (p.nchunk > 1) {
            size_t n4 = (size_t)B * S * H * D / 4;
            int blocks = (int)((n4 + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks), dim3(256), 0, st, dkp, (T*)dk, n4, n4, p.nchunk);
            hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks), dim3(256), 0, st, dvp, (T*)dv, n4, n4, p.nchunk);
            MMGL_CHECK_LAUNCH("xattn_bwd_reduce");
        }
        return MMGL_OK;
    }
    if constexpr (sizeof(T) == 2 && D <= 64 && NSB <= 4) {
        if (use_fused_bwd()) {                                // one pass over Q / dO (p.nsg == 1 here: all keys in one wave)
            constexpr int LDT = C::DPAD + 16, LDP = C::SPAD + 16;
            const size_t lds = sizeof(bf16) * (C::RMIMG + C::ROWIMG + 2 * 32 * LDT + 2 * 32 * LDP) + C::SPAD;
            auto kern = xattn_bwd_fused_kernel<D, NSB>;
            int rc = set_lds(kern, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(kern, dim3(B * H * p.nchunk), dim3(64), lds, st, (const bf16*)dout, (const bf16*)q, (const bf16*)k,
                               (const bf16*)v, lse, valid, (bf16*)dq, (bf16*)dk, (bf16*)dv, dkp, dvp, B, H, T_, S, p.rows_per_chunk,
                               p.nchunk);
            MMGL_CHECK_LAUNCH("xattn_bwd_fused");
            if (p.nchunk > 1) {
                size_t n4 = (size_t)B * S * H * D / 4;
                int blocks = (int)((n4 + 255) / 256);
                if (blocks > 2048) blocks = 2048;
                hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks), dim3(256), 0, st, dkp, (T*)dk, n4, n4, p.nchunk);
                hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks), dim3(256), 0, st, dvp, (T*)dv, n4, n4, p.nchunk);
                MMGL_CHECK_LAUNCH("xattn_bwd_reduce");
            }
            return MMGL_OK;
        }
    }
    {
        int rpw, nchunk;
        fwd_geometry(B, H, T_, 16 * C::QT, rpw, nchunk);
        size_t lds = sizeof(T) * (C::ROWIMG + (C::TIMG ? C::RMIMG : C::ROWIMG)) + C::SPAD;
        auto kern = xattn_bwd_dq_kernel<T, D, NSB>;
        int rc = set_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(B * H * nchunk), dim3(256), lds, st, (const T*)dout, (const T*)q, (const T*)k,
                           (const T*)v, lse, valid, (T*)dq, delta, B, H, T_, S, rpw, nchunk);
        MMGL_CHECK_LAUNCH("xattn_bwd_dq");
    }
    {
        typedef XC<T, D, 2> C2;
        constexpr int DLD = C2::DPAD + (sizeof(T) == 2 ? 8 : 4);
        size_t lds = sizeof(T) * (2 * C2::ROWIMG + 2 * 32 * DLD);
        auto kern = xattn_bwd_dkv_kernel<T, D>;
        int rc = set_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(B * H * p.nsg * p.nchunk), dim3(64), lds, st, (const T*)dout, (const T*)q,
                           (const T*)k, (const T*)v, lse, delta, valid, dkp, dvp, B, H, T_, S, p.nsg, p.rows_per_chunk,
                           p.nchunk);
        MMGL_CHECK_LAUNCH("xattn_bwd_dkv");
    }
    {
        size_t n4 = (size_t)B * S * H * D / 4;
        int blocks = (int)((n4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks), dim3(256), 0, st, dkp, (T*)dk, n4, n4, p.nchunk);
        hipLaunchKernelGGL(reduce_partials_kernel<T>, dim3(blocks), dim3(256), 0, st, dvp, (T*)dv, n4, n4, p.nchunk);
        MMGL_CHECK_LAUNCH("xattn_bwd_reduce");
    }
    return MMGL_OK;
}

int check_shape(const char* who, int B, int H, int T, int S, int D, int dtype) {
    MMGL_CHECK_ARG(B > 0 && H > 0 && T > 0 && S > 0, "%s: B,H,T,S must be positive (got %d,%d,%d,%d)", who, B, H, T, S);
    MMGL_CHECK_ARG(dtype == MMGL_F32 || dtype == MMGL_BF16, "%s: dtype must be MMGL_F32 or MMGL_BF16", who);
    if (!(D == 16 || D == 32 || D == 64 || D == 128))
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: head_dim %d not in {16,32,64,128}", who, D);
    if (S > 256) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: %d neighbor key tokens > 256 (single-pass kernel)", who, S);
    return MMGL_OK;
}

#define DISPATCH_NSB(FN, T, D, ...)                                    \
    do {                                                               \
        if (S <= 32) return FN<T, D, 2>(__VA_ARGS__);                  \
        if (S <= 64) return FN<T, D, 4>(__VA_ARGS__);                  \
        if (S <= 128) return FN<T, D, 8>(__VA_ARGS__);                 \
        return FN<T, D, 16>(__VA_ARGS__);                              \
    } while (0)
#define DISPATCH_D(FN, T, ...)                                         \
    do {                                                               \
        switch (D) {                                                   \
            case 16: DISPATCH_NSB(FN, T, 16, __VA_ARGS__);             \
            case 32: DISPATCH_NSB(FN, T, 32, __VA_ARGS__);             \
            case 64: DISPATCH_NSB(FN, T, 64, __VA_ARGS__);             \
            default: DISPATCH_NSB(FN, T, 128, __VA_ARGS__);            \
        }                                                              \
    } while (0)

}  // namespace

extern "C" int mmgl_xattn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out,
                              float* lse, int B, int H, int T, int S, int D, int dtype, void* stream) {
    int rc = check_shape("mmgl_xattn_fwd", B, H, T, S, D, dtype);
    if (rc) return rc;
    MMGL_CHECK_ARG(q && k && v && key_valid && out && lse, "mmgl_xattn_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) DISPATCH_D(launch_fwd, bf16, q, k, v, key_valid, out, lse, B, H, T, S, st);
    DISPATCH_D(launch_fwd, float, q, k, v, key_valid, out, lse, B, H, T, S, st);
}

extern "C" size_t mmgl_xattn_bwd_workspace(int B, int H, int T, int S, int D) {
    if (B <= 0 || H <= 0 || T <= 0 || S <= 0 || D <= 0) return 0;
    const size_t a = bwd_plan(B, H, T, S, D, 2).total, b = bwd_plan(B, H, T, S, D, 4).total;     // the entry point does not know the dtype
    return a > b ? a : b;
}

extern "C" int mmgl_xattn_bwd(const void* dout, const void* q, const void* k, const void* v, const float* lse,
                              const uint8_t* key_valid, void* dq, void* dk, void* dv, void* workspace,
                              size_t workspace_bytes, int B, int H, int T, int S, int D, int dtype, void* stream) {
    int rc = check_shape("mmgl_xattn_bwd", B, H, T, S, D, dtype);
    if (rc) return rc;
    MMGL_CHECK_ARG(dout && q && k && v && lse && key_valid && dq && dk && dv && workspace, "mmgl_xattn_bwd: null pointer");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_xattn_bwd_workspace(B, H, T, S, D), "mmgl_xattn_bwd: workspace %zu B < required %zu B",
                   workspace_bytes, mmgl_xattn_bwd_workspace(B, H, T, S, D));
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    if (dtype == MMGL_BF16) DISPATCH_D(launch_bwd, bf16, dout, q, k, v, lse, key_valid, dq, dk, dv, ws, B, H, T, S, st);
    DISPATCH_D(launch_bwd, float, dout, q, k, v, lse, key_valid, dq, dk, dv, ws, B, H, T, S, st);
}
