// Causal self-attention of the (frozen) decoder layers for gfx950, flash style:  O = softmax(mask(Q K^T)) V  with
// mask = causal AND key-valid, exactly the softmax the reference's additive finfo.min mask + clamp produces
// (model/modelling_cross_attention.py:51-79, 206-235, 455-476) whenever every query row keeps at least one key
// (the caller guarantees key 0 is valid: sequences are right-padded).  SURVEY.md 8(f) row 2.
//
// Same lane geometry as xattn.hip (swapped products, lane = one query row, masks as the MFMA C-input, P never moves
// between lanes, V^T / K^T fragments via ds_read_b64_tr_b16), plus the online-softmax loop over 64-key tiles that a
// 640..2176-key sequence needs.  Backward = flash backward: delta = rowsum(dO*O) pre-pass, a dQ kernel (loop over key
// tiles per query tile) and a dK/dV kernel (loop over query tiles per key tile, no atomics, no global partials).
#include "attn_common.h"
#include "selfattn32.h"
#include <type_traits>

namespace {

constexpr int KT = 64;            // keys per tile (NSB = 4)

template <typename T, int D> struct SC : XC<T, D, 4, 1> {};

// per-tile additive key bias in accumulator layout from the staged valid bytes (0 valid, -inf masked/absent).  Returns true
// (wave-uniform) when every key of the tile is valid: the bias is all zeros and the caller's masking code can be skipped.
template <typename C> __device__ __forceinline__ bool tile_bias(const uint8_t* vld, int g, f32x4 (&bias)[4]) {
    uint32_t w[4];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) w[sb] = *(const uint32_t*)(vld + sb * 16 + g * 4);       // staged as exactly 0 / 1 per key
    const bool lane_all = (w[0] & w[1] & w[2] & w[3]) == 0x01010101u;
    if (__builtin_amdgcn_ballot_w64(!lane_all) == 0ull) {
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) bias[sb] = vzero<f32x4>();
        return true;
    }
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[sb][r] = ((w[sb] >> (8 * r)) & 0xffu) ? 0.f : -INFINITY;
    return false;
}

__device__ __forceinline__ f32x4 vmax4(const f32x4& a, const f32x4& b) { return __builtin_elementwise_max(a, b); }

// One online-softmax step of a query row held as 16 scores per lane (4 key blocks x 4): new running max, P = exp2(...) in
// place, the lane's partial row sum (kept as a 4-vector: packed adds, folded once at the end), the output accumulators
// rescaled, P packed as the two B-operand fragments of the P.V product.  Written on 4-vectors so that hipcc emits the packed
// fp32 forms (v_pk_fma / v_pk_add / v_pk_mul) and v_max3 chains: about a third fewer VALU instructions than the scalar form,
// in kernels whose vector ALU is busier than their matrix pipe.
template <typename T, int NDB>
__device__ __forceinline__ void online_softmax_row(f32x4 (&s)[4], float& m, f32x4& lsum, f32x4 (&o)[NDB], typename Elem<T>::v8 (&pf)[2],
                                                   bool guard_empty) {
    const f32x4 m4 = vmax4(vmax4(vmax4(s[0], s[1]), s[2]), s[3]);
    float tm = fmaxf(fmaxf(fmaxf(m4[0], m4[1]), m4[2]), m4[3]);
    tm = xg_max(tm);
    const float mn = fmaxf(m, tm);
    const float mn2 = (guard_empty && mn == -INFINITY) ? 0.f : mn * LOG2E;      // row without any allowed key yet: keep p = 0
    const float alpha = __builtin_amdgcn_exp2f(m * LOG2E - mn2);                // m = -inf -> 0
    m = mn;
    const f32x4 nm = {-mn2, -mn2, -mn2, -mn2};
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        const f32x4 e = s[sb] * LOG2E + nm;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[sb][r] = __builtin_amdgcn_exp2f(e[r]);
    }
    lsum = lsum * alpha + ((s[0] + s[1]) + (s[2] + s[3]));
#pragma unroll
    for (int db = 0; db < NDB; ++db) o[db] *= alpha;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) pf[ks] = pack8<T>(s[2 * ks], s[2 * ks + 1]);
}

// Register-staged K / V tile (T14 split: the next tile's global loads are issued before the current tile's MFMAs and
// written to LDS after them, so the HBM/L2 latency of a 64-key tile hides under compute).
// "These registers are complete": an empty asm that reads them, placed ahead of a loop.  A load issued before a loop and first used
// inside it stays pending in hipcc's waitcnt scoreboard at the loop header, so every iteration waits with a count that means
// "the prefetch I just issued as well" -- the next tile's loads drained in the middle of the current tile's MFMAs.
template <typename V> __device__ __forceinline__ void loaded(const V& v) {
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass would check the "v" constraint against x86 register classes)
    asm volatile("" ::"v"(v));
#endif
}

template <typename T, typename C, bool K_ROWMAJOR, bool V_ROWMAJOR> struct TileStage {
    typedef typename Elem<T>::v8 v8;
    static constexpr int N = C::SPAD * C::CPR / 256;          // chunks per thread per operand (256 threads)
    v8 kr[N], vr[N];
    uint8_t vb;
    bool vin;
    __device__ __forceinline__ void load(const T* kbase, const T* vbase, const uint8_t* valid_row, size_t row_stride, int s0, int T_) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * 256, s = id / C::CPR, c = id % C::CPR;
            const bool ok = (s0 + s < T_) && (c * 8 < C::D);
            kr[i] = ok ? *(const v8*)(kbase + (size_t)(s0 + s) * row_stride + c * 8) : vzero<v8>();
            vr[i] = ok ? *(const v8*)(vbase + (size_t)(s0 + s) * row_stride + c * 8) : vzero<v8>();
        }
        vin = threadIdx.x < KT && s0 + (int)threadIdx.x < T_;
        vb = (vin && valid_row) ? valid_row[s0 + threadIdx.x] : (uint8_t)1;
    }
    // branch-free form: rows past the end of the sequence / padding channels fall outside the descriptor and read as 0
    __device__ __forceinline__ void loadb(__amdgpu_buffer_rsrc_t rk, __amdgpu_buffer_rsrc_t rv, uint32_t row_bytes,
                                          const uint8_t* valid_row, int s0, int T_) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * 256, s = id / C::CPR, c = id % C::CPR;
            const uint32_t off = row_off<T, C>(s0 + s, row_bytes, c * 8);
            kr[i] = buf_load8<T>(rk, off);
            vr[i] = buf_load8<T>(rv, off);
        }
        // the mask byte is only LOADED here: any arithmetic on it (even the in-range select) would make hipcc wait for it -- the
        // youngest load of the tile, i.e. vmcnt(0) -- right behind the prefetch instead of after this tile's MFMAs
        const int sv = s0 + (int)(threadIdx.x & (KT - 1));
        vb = valid_row ? valid_row[min(sv, T_ - 1)] : (uint8_t)1;
        vin = sv < T_;
    }
    __device__ __forceinline__ void store(T* Kimg, T* Vimg, uint8_t* vld) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int id = threadIdx.x + i * 256, s = id / C::CPR, c = id % C::CPR;
            if constexpr (K_ROWMAJOR) *(v8*)(Kimg + s * C::LD + c * 8) = kr[i];
            else *(v8*)(Kimg + rf_idx<C>(s >> 4, c >> 2, (s & 15) + 16 * (c & 3))) = kr[i];
            if constexpr (V_ROWMAJOR) *(v8*)(Vimg + s * C::LD + c * 8) = vr[i];
            else *(v8*)(Vimg + rf_idx<C>(s >> 4, c >> 2, (s & 15) + 16 * (c & 3))) = vr[i];
        }
        if (threadIdx.x < KT) vld[threadIdx.x] = (vin && vb) ? (uint8_t)1 : (uint8_t)0;
    }
};

// ============================================================================================ forward
template <typename T, int D>
__global__ __launch_bounds__(256) void selfattn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                           const T* __restrict__ v, const uint8_t* __restrict__ valid,
                                                           T* __restrict__ out, float* __restrict__ lse, int B, int H,
                                                           int T_, int nqb, int ldq, int P, int ldk) {
    // P > 0: keys / values carry P always-visible prefix rows in front of the T causal ones (peft prefix tuning: a learned
    // per-layer key/value prefix); k, v: [B, P + T, ldk], valid: [B, P + T]; key s is visible to query t iff s <= t + P
    typedef SC<T, D> C;
    typedef typename Elem<T>::v8 v8;
    constexpr int TILE = 16 * C::QT, QB = 4 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VIMG = C::TIMG ? C::RMIMG : C::ROWIMG;
    constexpr int BUFB = (int)(sizeof(T) * (C::ROWIMG + VIMG)) + KT;             // one K/V tile buffer (multiple of 16 B)
    auto Kimg = [&](int i) { return (T*)(smem + i * BUFB); };
    auto Vimg = [&](int i) { return (T*)(smem + i * BUFB) + C::ROWIMG; };
    auto Vld = [&](int i) { return (uint8_t*)(smem + i * BUFB) + sizeof(T) * (C::ROWIMG + VIMG); };

    const int vid = xcd_remap(blockIdx.x, B * H * nqb);
    const int bh = vid / nqb, qblk = nqb - 1 - vid % nqb;        // longest (most key tiles) first
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int t0 = qblk * QB + wave * TILE;
    const int Tk = T_ + P;
    const int nkt = (min(T_, (qblk + 1) * QB) + P + KT - 1) / KT;

    // q rows are ldq, k / v rows ldk elements apart (3 H D when they are column slices of one fused-QKV GEMM output)
    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T)), rbq = (uint32_t)(ldq * sizeof(T)), rbk = (uint32_t)(ldk * sizeof(T));
    const uint32_t slab = (uint32_t)(((size_t)(T_ - 1) * HD + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)b * T_ * ldq + h * D, (uint32_t)(((size_t)(T_ - 1) * ldq + D) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rl = make_rsrc(lse + (size_t)bh * T_, (uint32_t)(T_ * sizeof(float)));
    const uint32_t slabk = (uint32_t)(((size_t)(Tk - 1) * ldk + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(k + (size_t)b * Tk * ldk + h * D, slabk);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(v + (size_t)b * Tk * ldk + h * D, slabk);

    v8 qf[C::QT][C::NDC];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) qf[qt][dc] = buf_load8<T>(rq, row_off<T, C>(t0 + qt * 16 + x, rbq, dc * 32 + g * 8));

    float m[C::QT];
    f32x4 l[C::QT];
    f32x4 oacc[C::QT][C::NDB];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        m[qt] = -INFINITY;
        l[qt] = vzero<f32x4>();
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) oacc[qt][db] = vzero<f32x4>();
    }

    TileStage<T, C, false, C::TIMG> stg;
    stg.loadb(rk, rv, rbk, valid + (size_t)b * Tk, 0, Tk);
    stg.store(Kimg(0), Vimg(0), Vld(0));
    __syncthreads();
    // two LDS tile buffers, ONE barrier per key tile: tile j+1 is written into the other buffer after this wave's MFMAs of
    // tile j; the barrier at the end of iteration j both publishes it and proves every wave is done reading tile j-1's buffer.
    for (int j = 0; j < nkt; ++j) {
        const int s0 = j * KT;
        const T* Kf = Kimg(j & 1);
        const T* Vi = Vimg(j & 1);
        const uint8_t* vld = Vld(j & 1);
        if (j + 1 < nkt) stg.loadb(rk, rv, rbk, valid + (size_t)b * Tk, s0 + KT, Tk);
        if (s0 <= t0 + P + TILE - 1) {                        // else: tile entirely above this wave's diagonal (wave-uniform)

        f32x4 bias[4];
        tile_bias<C>(vld, g, bias);
        f32x4 sacc[C::QT][4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) sacc[qt][sb] = bias[sb];
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                const v8 kf = *(const v8*)(Kf + rf_idx<C>(sb, dc, lane));
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(sacc[qt][sb], kf, qf[qt][dc]);
            }
        }
        const bool diag = s0 + KT - 1 > t0 + P;               // some (key, row) pair of this wave violates s <= t + P
        v8 pf[C::QT][2];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt) {
            const int t = t0 + qt * 16 + x + P;
            if (diag) {
#pragma unroll
                for (int sb = 0; sb < 4; ++sb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (s0 + sb * 16 + g * 4 + r > t) sacc[qt][sb][r] = -INFINITY;
            }
            online_softmax_row<T, C::NDB>(sacc[qt], m[qt], l[qt], oacc[qt], pf[qt], true);
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v8 vt;
                if constexpr (C::TIMG) vt = rm_tfrag_tr16<C>(Vi, db, ks, lane);
                else vt = load_tfrag<T, C>(Vi, Vi, db, ks, lane);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(oacc[qt][db], vt, pf[qt][ks]);
            }
        }
        if (j + 1 < nkt) stg.store(Kimg((j + 1) & 1), Vimg((j + 1) & 1), Vld((j + 1) & 1));
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int t = t0 + qt * 16 + x;
        const float lt = xg_sum((l[qt][0] + l[qt][1]) + (l[qt][2] + l[qt][3]));
        const float inv = __builtin_amdgcn_rcpf(lt);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) buf_store4<T>(ro, row_off<T, C>(t, row_bytes, db * 16 + g * 4), oacc[qt][db] * inv);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, m[qt] + __logf(lt)), rl, (g == 0) ? (uint32_t)t * 4u : OOB, 0, 0);
    }
}

// ============================================================================================ encoder forward (packed, bidirectional)
// The frozen neighbor encoders (RoBERTa / CLIP ViT; get_text_embs / get_visual_embs, modelling_cross_attention.py:978-1027)
// run forward-only over PACKED tokens: sequence i owns rows cu[i] .. cu[i+1]-1 of a [ntok, ld] buffer (q, k, v may be three
// column slices of one fused-QKV GEMM output, ld = 3 H D).  No padding token is ever loaded or multiplied; every key of a
// sequence is visible to every query (no causal mask), so all key tiles are full work except the ragged last one.
// `q_rows` limits the query rows computed per sequence (1 = only the CLS row, which is all the last layer has to produce).
template <typename T, int D>
__global__ __launch_bounds__(256) void encattn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, const int* __restrict__ cu,
                                                          T* __restrict__ out, int nseq, int H, int ld_in, int ld_out,
                                                          int nqb, int q_rows) {
    typedef SC<T, D> C;
    typedef typename Elem<T>::v8 v8;
    constexpr int TILE = 16 * C::QT, QB = 4 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VIMG = C::TIMG ? C::RMIMG : C::ROWIMG;
    constexpr int BUFB = (int)(sizeof(T) * (C::ROWIMG + VIMG)) + KT;
    auto Kimg = [&](int i) { return (T*)(smem + i * BUFB); };
    auto Vimg = [&](int i) { return (T*)(smem + i * BUFB) + C::ROWIMG; };
    auto Vld = [&](int i) { return (uint8_t*)(smem + i * BUFB) + sizeof(T) * (C::ROWIMG + VIMG); };

    const int vid = xcd_remap(blockIdx.x, nseq * H * nqb);
    const int sh = vid / nqb, qblk = vid % nqb;
    const int seq = sh / H, h = sh % H;
    const int start = cu[seq], len = cu[seq + 1] - start;
    const int qlen = min(len, q_rows);
    if (qblk * QB >= qlen) return;                            // workgroup-uniform: before any barrier
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int t0 = qblk * QB + wave * TILE;
    const int nkt = (len + KT - 1) / KT;

    const uint32_t rb_in = (uint32_t)(ld_in * sizeof(T)), rb_out = (uint32_t)(ld_out * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)start * ld_in + h * D, (uint32_t)(((size_t)(qlen - 1) * ld_in + D) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)start * ld_out + h * D, (uint32_t)(((size_t)(qlen - 1) * ld_out + D) * sizeof(T)));
    const uint32_t slabk = (uint32_t)(((size_t)(len - 1) * ld_in + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(k + (size_t)start * ld_in + h * D, slabk);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(v + (size_t)start * ld_in + h * D, slabk);

    v8 qf[C::QT][C::NDC];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) qf[qt][dc] = buf_load8<T>(rq, row_off<T, C>(t0 + qt * 16 + x, rb_in, dc * 32 + g * 8));

    float m[C::QT];
    f32x4 l[C::QT];
    f32x4 oacc[C::QT][C::NDB];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        m[qt] = -INFINITY;
        l[qt] = vzero<f32x4>();
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) oacc[qt][db] = vzero<f32x4>();
    }

    TileStage<T, C, false, C::TIMG> stg;
    stg.loadb(rk, rv, rb_in, nullptr, 0, len);
    stg.store(Kimg(0), Vimg(0), Vld(0));
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) loaded(qf[qt][dc]);
    __syncthreads();
    for (int j = 0; j < nkt; ++j) {                           // double-buffered tiles, one barrier each (see selfattn_fwd_kernel)
        const T* Kf = Kimg(j & 1);
        const T* Vi = Vimg(j & 1);
        const uint8_t* vld = Vld(j & 1);
        if (j + 1 < nkt) stg.loadb(rk, rv, rb_in, nullptr, (j + 1) * KT, len);
        f32x4 bias[4];
        tile_bias<C>(vld, g, bias);                           // -inf for the keys past the end of the sequence
        f32x4 sacc[C::QT][4];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
            for (int qt = 0; qt < C::QT; ++qt) sacc[qt][sb] = bias[sb];
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                const v8 kf = *(const v8*)(Kf + rf_idx<C>(sb, dc, lane));
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(sacc[qt][sb], kf, qf[qt][dc]);
            }
        }
        v8 pf[C::QT][2];
#pragma unroll
        for (int qt = 0; qt < C::QT; ++qt)                    // the running max is finite from tile 0 on: every tile holds >= 1 real key
            online_softmax_row<T, C::NDB>(sacc[qt], m[qt], l[qt], oacc[qt], pf[qt], false);
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v8 vt;
                if constexpr (C::TIMG) vt = rm_tfrag_tr16<C>(Vi, db, ks, lane);
                else vt = load_tfrag<T, C>(Vi, Vi, db, ks, lane);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(oacc[qt][db], vt, pf[qt][ks]);
            }
        if (j + 1 < nkt) stg.store(Kimg((j + 1) & 1), Vimg((j + 1) & 1), Vld((j + 1) & 1));
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int t = t0 + qt * 16 + x;
        const float inv = __builtin_amdgcn_rcpf(xg_sum((l[qt][0] + l[qt][1]) + (l[qt][2] + l[qt][3])));
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) buf_store4<T>(ro, row_off<T, C>(t, rb_out, db * 16 + g * 4), oacc[qt][db] * inv);
    }
}

// ============================================================================================ delta = rowsum(dO * O) per head
template <typename T, int D>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ dout, const T* __restrict__ out, float* __restrict__ delta,
                                                     int B, int H, int T_) {
    typedef typename Elem<T>::v8 v8;
    // one thread = one (b, t, h, 8-channel chunk); D/8 consecutive threads share a head -> shuffle reduce
    constexpr int CPH = D / 8;
    const size_t total = (size_t)B * T_ * H * CPH;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (total + 63) / 64 * 64; i += (size_t)gridDim.x * 256) {
        float s = 0.f;
        if (i < total) {
            const v8 a = *(const v8*)(dout + i * 8), c = *(const v8*)(out + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)c[e];
        }
#pragma unroll
        for (int o = CPH / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (i < total && (i % CPH) == 0) {
            const size_t bth = i / CPH;                      // (b*T + t)*H + h
            const int h = (int)(bth % H);
            const size_t bt = bth / H;
            const int t = (int)(bt % T_), b = (int)(bt / T_);
            delta[((size_t)b * H + h) * T_ + t] = s;
        }
    }
}

// ============================================================================================ backward: dQ
template <typename T, int D>
__global__ __launch_bounds__(256) void selfattn_bwd_dq_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                              const T* __restrict__ k, const T* __restrict__ v,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              const uint8_t* __restrict__ valid, T* __restrict__ dq, int B, int H,
                                                              int T_, int nqb, int ldq, int ldg, int P, int ldk) {
    typedef XC<T, D, 4, 2> C;
    typedef typename Elem<T>::v8 v8;
    constexpr int TILE = 16 * C::QT, QB = 4 * TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KIMG = C::TIMG ? C::RMIMG : C::ROWIMG;      // K: row-major (bf16) / row image (f32)
    constexpr int BUFB = (int)(sizeof(T) * (KIMG + C::ROWIMG)) + KT;
    auto Kimg = [&](int i) { return (T*)(smem + i * BUFB); };
    auto Vimg = [&](int i) { return (T*)(smem + i * BUFB) + KIMG; };
    auto Vld = [&](int i) { return (uint8_t*)(smem + i * BUFB) + sizeof(T) * (KIMG + C::ROWIMG); };

    const int vid = xcd_remap(blockIdx.x, B * H * nqb);
    const int bh = vid / nqb, qblk = nqb - 1 - vid % nqb;
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int t0 = qblk * QB + wave * TILE;
    const int Tk = T_ + P;
    const int nkt = (min(T_, (qblk + 1) * QB) + P + KT - 1) / KT;

    const uint32_t row_bytes = (uint32_t)(HD * sizeof(T)), rbq = (uint32_t)(ldq * sizeof(T)), rbg = (uint32_t)(ldg * sizeof(T)), rbk = (uint32_t)(ldk * sizeof(T));
    const uint32_t slab = (uint32_t)(((size_t)(T_ - 1) * HD + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rq = make_rsrc(q + (size_t)b * T_ * ldq + h * D, (uint32_t)(((size_t)(T_ - 1) * ldq + D) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(dout + (size_t)b * T_ * HD + h * D, slab);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dq + (size_t)b * T_ * ldg + h * D, (uint32_t)(((size_t)(T_ - 1) * ldg + D) * sizeof(T)));
    const uint32_t slabk = (uint32_t)(((size_t)(Tk - 1) * ldk + D) * sizeof(T));
    const __amdgpu_buffer_rsrc_t rk = make_rsrc(k + (size_t)b * Tk * ldk + h * D, slabk);
    const __amdgpu_buffer_rsrc_t rv = make_rsrc(v + (size_t)b * Tk * ldk + h * D, slabk);

    v8 qf[C::QT][C::NDC], gf[C::QT][C::NDC];
    float lse2[C::QT], dlt[C::QT];
    // the row statistics are only LOADED here (clamped address, no select, no scaling): arithmetic right behind the load made
    // hipcc wait for it before the Q / dO / first K,V tile loads were even issued -- two extra memory round trips per workgroup
    float lraw[C::QT], draw[C::QT];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int t = t0 + qt * 16 + x;
        lraw[qt] = lse[(size_t)bh * T_ + min(t, T_ - 1)];
        draw[qt] = delta[(size_t)bh * T_ + min(t, T_ - 1)];
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) {
            qf[qt][dc] = buf_load8<T>(rq, row_off<T, C>(t, rbq, dc * 32 + g * 8));
            gf[qt][dc] = buf_load8<T>(rg, row_off<T, C>(t, row_bytes, dc * 32 + g * 8));
        }
    }
    f32x4 acc[C::QT][C::NDB];
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt)
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) acc[qt][db] = vzero<f32x4>();

    TileStage<T, C, C::TIMG, false> stg;
    stg.loadb(rk, rv, rbk, valid + (size_t)b * Tk, 0, Tk);
    stg.store(Kimg(0), Vimg(0), Vld(0));
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const bool in = t0 + qt * 16 + x < T_;
        lse2[qt] = in ? lraw[qt] * LOG2E : 0.f;
        dlt[qt] = in ? draw[qt] : 0.f;
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) { loaded(qf[qt][dc]); loaded(gf[qt][dc]); }
    }
    __syncthreads();
    for (int j = 0; j < nkt; ++j) {                           // double-buffered tiles, one barrier each (see selfattn_fwd_kernel)
        const int s0 = j * KT;
        const T* Ki = Kimg(j & 1);
        const T* Vf = Vimg(j & 1);
        const uint8_t* vld = Vld(j & 1);
        if (j + 1 < nkt) stg.loadb(rk, rv, rbk, valid + (size_t)b * Tk, s0 + KT, Tk);
        if (s0 <= t0 + P + TILE - 1) {

        f32x4 bias[4];
        tile_bias<C>(vld, g, bias);
        const bool diag = s0 + KT - 1 > t0 + P;
        v8 dsf[C::QT][2];
        {
            // VALU diet: -delta enters as the C-input of the dP product (acc = dP - delta), the exp argument and dS are packed
            // 4-vector ops, the causal select exists only on diagonal tiles
            f32x4 sacc[C::QT][4], pacc[C::QT][4];
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) {
                    sacc[qt][sb] = bias[sb];
                    pacc[qt][sb] = f32x4{-dlt[qt], -dlt[qt], -dlt[qt], -dlt[qt]};
                }
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
                const int t = t0 + qt * 16 + x + P;
                const f32x4 nl = {-lse2[qt], -lse2[qt], -lse2[qt], -lse2[qt]};
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) {
                    f32x4 e = sacc[qt][sb] * LOG2E + nl;
                    if (diag) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) e[r] = (s0 + sb * 16 + g * 4 + r > t) ? -INFINITY : e[r];
                    }
                    f32x4 p4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) p4[r] = __builtin_amdgcn_exp2f(e[r]);
                    sacc[qt][sb] = p4 * pacc[qt][sb];
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) dsf[qt][ks] = pack8<T>(sacc[qt][2 * ks], sacc[qt][2 * ks + 1]);
            }
        }
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v8 kt;
                if constexpr (C::TIMG) kt = rm_tfrag_tr16<C>(Ki, db, ks, lane);
                else kt = load_tfrag<T, C>(Ki, Ki, db, ks, lane);
#pragma unroll
                for (int qt = 0; qt < C::QT; ++qt) mma16(acc[qt][db], kt, dsf[qt][ks]);
            }
        }
        if (j + 1 < nkt) stg.store(Kimg((j + 1) & 1), Vimg((j + 1) & 1), Vld((j + 1) & 1));
        __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < C::QT; ++qt) {
        const int t = t0 + qt * 16 + x;
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) buf_store4<T>(rd, row_off<T, C>(t, rbg, db * 16 + g * 4), acc[qt][db]);
    }
}

// ============================================================================================ backward: dK, dV
// Workgroup = (b, h, 64-key tile); wave w: key half (w>>1) of 32 keys, query tiles of parity (w&1).  Non-swapped products:
// lane = key column, P / dS leave the accumulators in B-operand layout for the contraction over t; Q^T / dO^T come
// from a wave-private row-major LDS tile via tr16 (bf16) or scalar gathers (f32).  The two parity waves of a key half
// are folded through LDS at the end: every dK / dV element is written exactly once (deterministic, no partials).
template <typename T, int D, int PAR>
__global__ __launch_bounds__(128 * PAR) void selfattn_bwd_dkv_kernel(const T* __restrict__ dout, const T* __restrict__ q,
                                                               const T* __restrict__ k, const T* __restrict__ v,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const uint8_t* __restrict__ valid, T* __restrict__ dk,
                                                               T* __restrict__ dv, int B, int H, int T_, int nkb, int ldq, int ldg,
                                                               int P, int ldk) {
    typedef XC<T, D, 2> C;                         // one wave's key group: 32 keys = 2 blocks
    const int Tk = T_ + P;                         // keys: P always-visible prefix rows, then the T causal ones
    typedef typename Elem<T>::v8 v8;
    constexpr int LDT = C::DPAD + 16;              // row stride of the wave-private tiles (elements)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Kb = (T*)smem;                              // [2 halves][ROWIMG]
    T* Vb = Kb + 2 * C::ROWIMG;
    T* tiles = Vb + 2 * C::ROWIMG;                 // [4 waves][2 (Q, dO)][32 * LDT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int half = wave / PAR, par = wave % PAR;           // PAR = 1: two waves, one per key half (fp32 D = 128: LDS budget)
    T* Qt = tiles + (size_t)wave * 2 * 32 * LDT;
    T* Gt = Qt + 32 * LDT;

    const int vid = xcd_remap(blockIdx.x, B * H * nkb);
    const int bh = vid / nkb, kblk = vid % nkb;                 // low key tiles (most query tiles) first
    const int b = bh / H, h = bh % H;
    const size_t HD = (size_t)H * D;
    const int s0 = kblk * KT + half * 32;
    const T* kb = k + ((size_t)b * Tk + kblk * KT) * ldk + h * D;
    const T* vb = v + ((size_t)b * Tk + kblk * KT) * ldk + h * D;

    // stage both halves' K / V row fragments (64 keys): image `half` holds keys [32 half, 32 half + 32)
    for (int hh = 0; hh < 2; ++hh) {
        stage_row_image<T, C>(Kb + hh * C::ROWIMG, kb + (size_t)hh * 32 * ldk, (size_t)ldk, Tk - (kblk * KT + hh * 32));
        stage_row_image<T, C>(Vb + hh * C::ROWIMG, vb + (size_t)hh * 32 * ldk, (size_t)ldk, Tk - (kblk * KT + hh * 32));
    }
    bool vs[2];
#pragma unroll
    for (int sbl = 0; sbl < 2; ++sbl) {
        const int s = s0 + sbl * 16 + x;
        vs[sbl] = s < Tk && valid[(size_t)b * Tk + s] != 0;
    }
    __syncthreads();
    const T* Kh = Kb + half * C::ROWIMG;
    const T* Vh = Vb + half * C::ROWIMG;

    f32x4 dva[C::NDB][2], dka[C::NDB][2];
#pragma unroll
    for (int db = 0; db < C::NDB; ++db)
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) { dva[db][sbl] = vzero<f32x4>(); dka[db][sbl] = vzero<f32x4>(); }

    const T* qb = q + (size_t)b * T_ * ldq + h * D;
    const T* gb = dout + (size_t)b * T_ * HD + h * D;
    const float* lb = lse + (size_t)bh * T_;
    const float* dlb = delta + (size_t)bh * T_;
    const int tfirst = (max(s0 - P, 0) / 32) * 32;              // first 32-row query tile that can see key s0 (visible iff s <= t + P)

    v8 qn[2][C::NDC], gn[2][C::NDC];                         // next tile's fragments, in flight during this tile's MFMAs
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int dc = 0; dc < C::NDC; ++dc) {
            const int t = tfirst + par * 32 + tb * 16 + x;
            qn[tb][dc] = load_qfrag<T, C>(qb, t, T_, (size_t)ldq, dc * 32 + g * 8);
            gn[tb][dc] = load_qfrag<T, C>(gb, t, T_, HD, dc * 32 + g * 8);
        }
    for (int t0 = tfirst + par * 32; t0 < T_; t0 += 32 * PAR) {
        v8 qa[2][C::NDC], ga[2][C::NDC];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int dc = 0; dc < C::NDC; ++dc) {
                qa[tb][dc] = qn[tb][dc];
                ga[tb][dc] = gn[tb][dc];
                const int tn = t0 + 32 * PAR + tb * 16 + x;
                qn[tb][dc] = load_qfrag<T, C>(qb, tn, T_, (size_t)ldq, dc * 32 + g * 8);
                gn[tb][dc] = load_qfrag<T, C>(gb, tn, T_, HD, dc * 32 + g * 8);
                *(v8*)(Qt + (tb * 16 + x) * LDT + dc * 32 + g * 8) = qa[tb][dc];
                *(v8*)(Gt + (tb * 16 + x) * LDT + dc * 32 + g * 8) = ga[tb][dc];
            }
        const bool diag = t0 + P < s0 + 31;                     // some (row, key) pair with key > row + P
        f32x4 pr[2][2], dsr[2][2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            float lt[4], dt[4];
            bool tv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + tb * 16 + g * 4 + r;
                tv[r] = t < T_;
                lt[r] = tv[r] ? lb[t] * LOG2E : 0.f;
                dt[r] = tv[r] ? dlb[t] : 0.f;
            }
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                f32x4 sa = vzero<f32x4>(), pa = vzero<f32x4>();
#pragma unroll
                for (int dc = 0; dc < C::NDC; ++dc) {
                    const v8 kf = *(const v8*)(Kh + rf_idx<C>(sbl, dc, lane));
                    const v8 vf = *(const v8*)(Vh + rf_idx<C>(sbl, dc, lane));
                    mma16(sa, qa[tb][dc], kf);
                    mma16(pa, ga[tb][dc], vf);
                }
                const int s = s0 + sbl * 16 + x;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + tb * 16 + g * 4 + r;
                    const bool ok = tv[r] && vs[sbl] && (!diag || s <= t + P);
                    const float p = ok ? __builtin_amdgcn_exp2f(fmaf(sa[r], LOG2E, -lt[r])) : 0.f;
                    pr[tb][sbl][r] = p;
                    dsr[tb][sbl][r] = p * (pa[r] - dt[r]);
                }
            }
        }
        v8 pB[2], dsB[2];
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) {
            pB[sbl] = pack8<T>(pr[0][sbl], pr[1][sbl]);
            dsB[sbl] = pack8<T>(dsr[0][sbl], dsr[1][sbl]);
        }
        // wave-private tiles: LDS ops of one wave execute in order; the fence only stops compiler reordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < C::NDB; ++db) {
            v8 gT, qT;
            if constexpr (sizeof(T) == 2) {
                typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
                const int i = lane & 15;
                const bf16* pg = (const bf16*)Gt + (4 * g + (i >> 2)) * LDT + db * 16 + (i & 3) * 4;
                const bf16* pq = (const bf16*)Qt + (4 * g + (i >> 2)) * LDT + db * 16 + (i & 3) * 4;
                const bf16x4 g0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pg);
                const bf16x4 g1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pg + 16 * LDT));
                const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)pq);
                const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(pq + 16 * LDT));
                gT = bf16x8{g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                qT = bf16x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int row = (e >> 2) * 16 + g * 4 + (e & 3);
                    gT[e] = Gt[row * LDT + db * 16 + x];
                    qT[e] = Qt[row * LDT + db * 16 + x];
                }
            }
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                mma16(dva[db][sbl], gT, pB[sbl]);
                mma16(dka[db][sbl], qT, dsB[sbl]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // fold the two query-parity waves of each key half through LDS (reuse the tile region), then store
    __syncthreads();
    float* red = (float*)tiles;                                // [2 halves][NDB][2][64 lanes][4] floats x 2 (dk, dv)
    const int per = C::NDB * 2 * 64 * 4;
    if (PAR == 2 && par == 1) {
#pragma unroll
        for (int db = 0; db < C::NDB; ++db)
#pragma unroll
            for (int sbl = 0; sbl < 2; ++sbl) {
                *(f32x4*)(red + (size_t)half * 2 * per + ((db * 2 + sbl) * 64 + lane) * 4) = dka[db][sbl];
                *(f32x4*)(red + (size_t)half * 2 * per + per + ((db * 2 + sbl) * 64 + lane) * 4) = dva[db][sbl];
            }
    }
    __syncthreads();
    if (par == 0) {
#pragma unroll
        for (int sbl = 0; sbl < 2; ++sbl) {
            const int s = s0 + sbl * 16 + x;
            if (s < Tk) {
                const size_t off = ((size_t)b * Tk + s) * ldg + h * D + g * 4;
#pragma unroll
                for (int db = 0; db < C::NDB; ++db) {
                    f32x4 a = dka[db][sbl], c = dva[db][sbl];
                    if (PAR == 2) {
                        a += *(const f32x4*)(red + (size_t)half * 2 * per + ((db * 2 + sbl) * 64 + lane) * 4);
                        c += *(const f32x4*)(red + (size_t)half * 2 * per + per + ((db * 2 + sbl) * 64 + lane) * 4);
                    }
                    store4<T>(dk + off + db * 16, a);
                    store4<T>(dv + off + db * 16, c);
                }
            }
        }
    }
}

// ============================================================================================ host
template <typename K> int set_lds_sa(K kern, size_t bytes) {
    if (bytes > 160 * 1024) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "selfattn: needs %zu B of LDS (> 160 KiB)", bytes);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return MMGL_OK;
}

template <typename T, int D>
int sa_fwd(const void* q, const void* k, const void* v, const uint8_t* valid, void* out, float* lse, int B, int H, int T_,
           int ldq, int P, int ldk, hipStream_t st) {
    typedef SC<T, D> C;
    const int QB = 4 * 16 * C::QT, nqb = cdiv(T_, QB);
    const size_t lds = 2 * (sizeof(T) * (C::ROWIMG + (C::TIMG ? C::RMIMG : C::ROWIMG)) + KT);
    auto kern = selfattn_fwd_kernel<T, D>;
    int rc = set_lds_sa(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(B * H * nqb), dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, valid, (T*)out, lse, B, H,
                       T_, nqb, ldq, P, ldk);
    MMGL_CHECK_LAUNCH("selfattn_fwd");
    return MMGL_OK;
}

template <typename T, int D>
int sa_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse, const uint8_t* valid,
           void* dq, void* dk, void* dv, float* delta, int B, int H, int T_, int ldq, int ldg, int P, int ldk, int ldgk, hipStream_t st) {
    // ldq / ldg: row strides of q / dq; ldk / ldgk: of k, v / dk, dv ([B, P + T] rows); P prefix keys (0 = plain causal attention)
    {
        const size_t total = (size_t)B * T_ * H * (D / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL((rowdot_kernel<T, D>), dim3(blocks), dim3(256), 0, st, (const T*)dout, (const T*)out, delta, B, H, T_);
        MMGL_CHECK_LAUNCH("selfattn_rowdot");
    }
    {
        typedef XC<T, D, 4, 2> C;
        const int QB = 4 * 16 * C::QT, nqb = cdiv(T_, QB);
        const size_t lds = 2 * (sizeof(T) * ((C::TIMG ? C::RMIMG : C::ROWIMG) + C::ROWIMG) + KT);
        auto kern = selfattn_bwd_dq_kernel<T, D>;
        int rc = set_lds_sa(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3(B * H * nqb), dim3(256), lds, st, (const T*)dout, (const T*)q, (const T*)k, (const T*)v, lse,
                           delta, valid, (T*)dq, B, H, T_, nqb, ldq, ldg, P, ldk);
        MMGL_CHECK_LAUNCH("selfattn_bwd_dq");
    }
    {
        typedef XC<T, D, 2> C;
        constexpr int LDT = C::DPAD + 16;
        const int nkb = cdiv(T_ + P, KT);
        const size_t red = sizeof(float) * 2 * 2 * (C::NDB * 2 * 64 * 4);
        auto lds_for = [&](int par) {
            size_t tiles = sizeof(T) * (2 * par) * 2 * 32 * LDT;
            if (par == 2 && red > tiles) tiles = red;
            return sizeof(T) * 4 * C::ROWIMG + tiles;
        };
        if (lds_for(2) <= 160 * 1024) {
            auto kern = selfattn_bwd_dkv_kernel<T, D, 2>;
            int rc = set_lds_sa(kern, lds_for(2));
            if (rc) return rc;
            hipLaunchKernelGGL(kern, dim3(B * H * nkb), dim3(256), lds_for(2), st, (const T*)dout, (const T*)q, (const T*)k, (const T*)v,
                               lse, delta, valid, (T*)dk, (T*)dv, B, H, T_, nkb, ldq, ldgk, P, ldk);
        } else {
            auto kern = selfattn_bwd_dkv_kernel<T, D, 1>;
            int rc = set_lds_sa(kern, lds_for(1));
            if (rc) return rc;
            hipLaunchKernelGGL(kern, dim3(B * H * nkb), dim3(128), lds_for(1), st, (const T*)dout, (const T*)q, (const T*)k, (const T*)v,
                               lse, delta, valid, (T*)dk, (T*)dv, B, H, T_, nkb, ldq, ldgk, P, ldk);
        }
        MMGL_CHECK_LAUNCH("selfattn_bwd_dkv");
    }
    return MMGL_OK;
}

template <typename T, int D>
int enc_fwd(const void* q, const void* k, const void* v, const int* cu, void* out, int nseq, int H, int ld_in, int ld_out,
            int max_len, int q_rows, hipStream_t st) {
    typedef SC<T, D> C;
    const int QB = 4 * 16 * C::QT;
    const int nqb = cdiv(max_len < q_rows ? max_len : q_rows, QB);
    const size_t lds = 2 * (sizeof(T) * (C::ROWIMG + (C::TIMG ? C::RMIMG : C::ROWIMG)) + KT);
    auto kern = encattn_fwd_kernel<T, D>;
    int rc = set_lds_sa(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(nseq * H * nqb), dim3(256), lds, st, (const T*)q, (const T*)k, (const T*)v, cu, (T*)out, nseq, H,
                       ld_in, ld_out, nqb, q_rows);
    MMGL_CHECK_LAUNCH("encattn_fwd");
    return MMGL_OK;
}

int sa_check(const char* who, int B, int H, int T, int D, int dtype) {
    MMGL_CHECK_ARG(B > 0 && H > 0 && T > 0, "%s: B,H,T must be positive (got %d,%d,%d)", who, B, H, T);
    MMGL_CHECK_ARG(dtype == MMGL_F32 || dtype == MMGL_BF16, "%s: dtype must be MMGL_F32 or MMGL_BF16", who);
    if (!(D == 16 || D == 32 || D == 64 || D == 128)) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: head_dim %d not in {16,32,64,128}", who, D);
    return MMGL_OK;
}

#define SA_DISPATCH(FN, T, ...)                        \
    switch (D) {                                       \
        case 16: return FN<T, 16>(__VA_ARGS__);        \
        case 32: return FN<T, 32>(__VA_ARGS__);        \
        case 64: return FN<T, 64>(__VA_ARGS__);        \
        default: return FN<T, 128>(__VA_ARGS__);       \
    }

}  // namespace

static int sa_ld(const char* who, int& ld, int H, int D) {
    if (ld == 0) ld = H * D;
    MMGL_CHECK_ARG(ld >= H * D && ld % 8 == 0, "%s: row stride %d must be 0 (packed) or a multiple of 8 >= H*D = %d", who, ld, H * D);
    return MMGL_OK;
}

extern "C" int mmgl_selfattn_prefix_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out, float* lse,
                                        int B, int H, int T, int P, int D, int ld_q, int ld_kv, int dtype, void* stream) {
    int rc = sa_check("mmgl_selfattn_prefix_fwd", B, H, T, D, dtype);
    if (rc) return rc;
    MMGL_CHECK_ARG(q && k && v && key_valid && out && lse && P >= 0, "mmgl_selfattn_prefix_fwd: bad arguments");
    if ((rc = sa_ld("mmgl_selfattn_prefix_fwd", ld_q, H, D)) || (rc = sa_ld("mmgl_selfattn_prefix_fwd", ld_kv, H, D))) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16 && sa32_supported(D, T + P)) return sa32_fwd(q, k, v, key_valid, out, lse, B, H, T, P, D, ld_q, ld_kv, st);
    if (dtype == MMGL_BF16) { SA_DISPATCH(sa_fwd, bf16, q, k, v, key_valid, out, lse, B, H, T, ld_q, P, ld_kv, st) }
    SA_DISPATCH(sa_fwd, float, q, k, v, key_valid, out, lse, B, H, T, ld_q, P, ld_kv, st)
}

extern "C" int mmgl_selfattn_fwd(const void* q, const void* k, const void* v, const uint8_t* key_valid, void* out, float* lse,
                                 int B, int H, int T, int D, int ld_qkv, int dtype, void* stream) {
    return mmgl_selfattn_prefix_fwd(q, k, v, key_valid, out, lse, B, H, T, 0, D, ld_qkv, ld_qkv, dtype, stream);
}

extern "C" size_t mmgl_selfattn_bwd_workspace(int B, int H, int T) {
    if (B <= 0 || H <= 0 || T <= 0) return 0;
    return align_up((size_t)B * H * T * sizeof(float), 256);
}

extern "C" int mmgl_selfattn_prefix_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
                                        const uint8_t* key_valid, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                                        int B, int H, int T, int P, int D, int ld_q, int ld_kv, int ld_dq, int ld_dkv, int dtype, void* stream) {
    int rc = sa_check("mmgl_selfattn_prefix_bwd", B, H, T, D, dtype);
    if (rc) return rc;
    if ((rc = sa_ld("mmgl_selfattn_prefix_bwd", ld_q, H, D)) || (rc = sa_ld("mmgl_selfattn_prefix_bwd", ld_kv, H, D)) ||
        (rc = sa_ld("mmgl_selfattn_prefix_bwd", ld_dq, H, D)) || (rc = sa_ld("mmgl_selfattn_prefix_bwd", ld_dkv, H, D))) return rc;
    MMGL_CHECK_ARG(dout && q && k && v && out && lse && key_valid && dq && dk && dv && workspace && P >= 0, "mmgl_selfattn_prefix_bwd: bad arguments");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_selfattn_bwd_workspace(B, H, T), "mmgl_selfattn_prefix_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* delta = (float*)workspace;
    if (dtype == MMGL_BF16 && sa32_supported(D, T + P)) {
        // dQ (+ delta) kernel, then the dK / dV kernel (head_dim 128: one workgroup per CU, 304 registers; Llama shape 1743 -> 1289 us
        // against the 16x16 dK / dV kernel it replaced)
        return sa32_bwd(dout, q, k, v, out, lse, key_valid, dq, dk, dv, delta, B, H, T, P, D, ld_q, ld_kv, ld_dq, ld_dkv, 3, st);
    }
    if (dtype == MMGL_BF16) { SA_DISPATCH(sa_bwd, bf16, dout, q, k, v, out, lse, key_valid, dq, dk, dv, delta, B, H, T, ld_q, ld_dq, P, ld_kv, ld_dkv, st) }
    SA_DISPATCH(sa_bwd, float, dout, q, k, v, out, lse, key_valid, dq, dk, dv, delta, B, H, T, ld_q, ld_dq, P, ld_kv, ld_dkv, st)
}

extern "C" int mmgl_selfattn_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
                                 const uint8_t* key_valid, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes,
                                 int B, int H, int T, int D, int ld_qkv, int ld_dqkv, int dtype, void* stream) {
    return mmgl_selfattn_prefix_bwd(dout, q, k, v, out, lse, key_valid, dq, dk, dv, workspace, workspace_bytes, B, H, T, 0, D, ld_qkv,
                                    ld_qkv, ld_dqkv, ld_dqkv, dtype, stream);
}

extern "C" int mmgl_encattn_fwd(const void* q, const void* k, const void* v, const int32_t* cu_seqlens, void* out, int nseq,
                                int H, int D, int ld_in, int ld_out, int max_len, int q_rows, int dtype, void* stream) {
    int rc = sa_check("mmgl_encattn_fwd", nseq, H, max_len, D, dtype);
    if (rc) return rc;
    MMGL_CHECK_ARG(q && k && v && cu_seqlens && out, "mmgl_encattn_fwd: null pointer");
    MMGL_CHECK_ARG(q_rows > 0, "mmgl_encattn_fwd: q_rows must be positive (got %d)", q_rows);
    MMGL_CHECK_ARG(ld_in >= H * D && ld_out >= H * D && ld_in % 8 == 0 && ld_out % 8 == 0,
                   "mmgl_encattn_fwd: row strides (%d, %d) must be >= H*D = %d and multiples of 8 elements", ld_in, ld_out, H * D);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16 && sa32_supported(D, max_len)) return sa32_enc_fwd(q, k, v, cu_seqlens, out, nseq, H, D, ld_in, ld_out, max_len, q_rows, st);
    if (dtype == MMGL_BF16) { SA_DISPATCH(enc_fwd, bf16, q, k, v, cu_seqlens, out, nseq, H, ld_in, ld_out, max_len, q_rows, st) }
    SA_DISPATCH(enc_fwd, float, q, k, v, cu_seqlens, out, nseq, H, ld_in, ld_out, max_len, q_rows, st)
}
