// Device helpers shared by the attention kernels (xattn.hip: neighbor cross-attention; selfattn.hip: causal self-attention
// of the frozen LM layers): fragment geometry, LDS images, bounds-checked buffer access, cross-lane reductions.
#pragma once
#include "common.h"
#include <stdlib.h>

#ifndef MMGL_XATTN_HOIST_MAX
#define MMGL_XATTN_HOIST_MAX 16
#endif

namespace {

// QT = 16-row query tiles per iteration: 2 halves the LDS operand traffic per query row but doubles the live
// accumulators; budget = work_scale (1 fwd, 2 dq) * NSB * NDB 16x16 tiles, fp32 operands cost twice the registers.
template <typename T, int D_, int NSB_, int WORK_ = 1> struct XC {
    static constexpr int D = D_;
    static constexpr int NSB = NSB_;                 // 16-key blocks (S padded to NSB*16)
    static constexpr int NDC = (D + 31) / 32;        // 32-wide contraction chunks over d
    static constexpr int DPAD = NDC * 32;
    static constexpr int NDB = D / 16;               // 16-wide output channel blocks
    static constexpr int NKS = NSB / 2;              // 32-key contraction steps
    static constexpr int SPAD = NSB * 16;
    static constexpr int CPR = DPAD / 8;             // 8-element chunks per (padded) row
#ifdef MMGL_XATTN_QT_FORCE                           // (experiments: the forward kernel only -- the fused backward needs 32-row tiles)
    static constexpr int QT = (WORK_ == 1) ? MMGL_XATTN_QT_FORCE : ((WORK_ * NSB_ * (D_ / 16) * (int)(sizeof(T) / 2) <= 32) ? 2 : 1);
#else
    static constexpr int QT = (WORK_ * NSB_ * (D_ / 16) * (int)(sizeof(T) / 2) <= 32) ? 2 : 1;
#endif
    static constexpr bool TIMG = (sizeof(T) == 2);   // dedicated transposed image (bf16) vs gather (f32)
    // The K/V operand fragments are loop invariant, so the compiler hoists them into registers when it may
    // (register-resident K/V, no LDS traffic in the loop).  Past this budget that spills: re-read LDS instead.
    static constexpr bool HOIST = (WORK_ * NSB_ * (D_ / 16) * (int)(sizeof(T) / 2) <= MMGL_XATTN_HOIST_MAX);
    static constexpr int ROWIMG = SPAD * DPAD;       // elements
    static constexpr int TIMGSZ = D * SPAD;          // elements
    // row-major padded image (bf16): row stride = 2*DPAD + 32 bytes, so the 8 rows two 16-lane groups touch in one
    // ds_read_b64_tr_b16 cycle fall on 8 distinct 32-byte bank slots (conflict free), and a row-fragment ds_read_b128
    // is at most 2-way conflicted.
    static constexpr int LD = DPAD + 16;
    static constexpr int RMIMG = SPAD * LD;
};

template <typename C> __device__ __forceinline__ int rf_idx(int sb, int dc, int lane) {
    return ((sb * C::NDC + dc) * 64 + lane) * 8;
}
template <typename C> __device__ __forceinline__ int tf_idx(int db, int ks, int lane) {
    return ((db * C::NKS + ks) * 64 + lane) * 8;
}

// Stage X[b, 0..S, h*D .. h*D+D] into a row image (zero padded to SPAD x DPAD).
template <typename T, typename C>
__device__ __forceinline__ void stage_row_image(T* img, const T* base, int row_stride, int S) {
    typedef typename Elem<T>::v8 v8;
    for (int i = threadIdx.x; i < C::SPAD * C::CPR; i += blockDim.x) {
        const int s = i / C::CPR, c = i % C::CPR;
        v8 val = vzero<v8>();
        if (s < S && c * 8 < C::D) val = *(const v8*)(base + (size_t)s * row_stride + c * 8);
        *(v8*)(img + rf_idx<C>(s >> 4, c >> 2, (s & 15) + 16 * (c & 3))) = val;
    }
}
// Stage the transposed / key-permuted image.
template <typename T, typename C>
__device__ __forceinline__ void stage_t_image(T* img, const T* base, int row_stride, int S) {
    typedef typename Elem<T>::v8 v8;
    for (int i = threadIdx.x; i < C::SPAD * (C::D / 8); i += blockDim.x) {
        const int s = i / (C::D / 8), c = i % (C::D / 8);
        v8 val = vzero<v8>();
        if (s < S) val = *(const v8*)(base + (size_t)s * row_stride + c * 8);
        const int ks = s >> 5, e = (((s & 31) >> 4) << 2) + (s & 3), g = (s & 15) >> 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = c * 8 + j;
            img[tf_idx<C>(d >> 4, ks, (d & 15) + 16 * g) + e] = val[j];
        }
    }
}
// Transposed fragment (db, ks) for this lane, either from the t image or gathered from the row image.
template <typename T, typename C>
__device__ __forceinline__ typename Elem<T>::v8 load_tfrag(const T* timg, const T* rimg, int db, int ks, int lane) {
    typedef typename Elem<T>::v8 v8;
    if constexpr (C::TIMG) {
        return *(const v8*)(timg + tf_idx<C>(db, ks, lane));
    } else {
        const int x = lane & 15, g = lane >> 4;
        const int d = db * 16 + x;
        const int dc = d >> 5, gg = (d & 31) >> 3, ee = d & 7;
        v8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int sb = 2 * ks + (e >> 2), xs = g * 4 + (e & 3);
            r[e] = rimg[rf_idx<C>(sb, dc, xs + 16 * gg) + ee];
        }
        return r;
    }
}

// Row-major padded image: X[s][0..DPAD) at img + s*LD.
template <typename T, typename C>
__device__ __forceinline__ void stage_rowmajor_image(T* img, const T* base, int row_stride, int S) {
    typedef typename Elem<T>::v8 v8;
    for (int i = threadIdx.x; i < C::SPAD * C::CPR; i += blockDim.x) {
        const int s = i / C::CPR, c = i % C::CPR;
        v8 val = vzero<v8>();
        if (s < S && c * 8 < C::D) val = *(const v8*)(base + (size_t)s * row_stride + c * 8);
        *(v8*)(img + s * C::LD + c * 8) = val;
    }
}
// A-operand row fragment (sb, dc) out of a row-major image.
template <typename T, typename C>
__device__ __forceinline__ typename Elem<T>::v8 rm_rowfrag(const T* img, int sb, int dc, int lane) {
    return *(const typename Elem<T>::v8*)(img + (sb * 16 + (lane & 15)) * C::LD + dc * 32 + (lane >> 4) * 8);
}
// Transposed fragment (db, ks) out of a row-major bf16 image with the LDS transpose read (ds_read_b64_tr_b16):
// in each 16-lane group, lane i supplies the address of 4 consecutive channels of key row 4g + (i>>2); the hardware
// hands lane x the 4 keys of channel x.  Two reads (keys +0, +16) give exactly the key order the score accumulator
// uses (s = 16(2ks + (e>>2)) + 4g + (e&3)), so no dedicated transposed image has to be built.
template <typename C>
__device__ __forceinline__ bf16x8 rm_tfrag_tr16(const bf16* img, int db, int ks, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int i = lane & 15, g = lane >> 4;
    const bf16* p = img + (ks * 32 + 4 * g + (i >> 2)) * C::LD + db * 16 + (i & 3) * 4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * C::LD));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

template <typename T, typename C>
__device__ __forceinline__ typename Elem<T>::v8 load_qfrag(const T* base, int t, int T_, size_t row_stride, int dcol) {
    typedef typename Elem<T>::v8 v8;
    if (t < T_ && dcol < C::D) return *(const v8*)(base + (size_t)t * row_stride + dcol);
    return vzero<v8>();
}

template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const f32x4& v) {
    *(bf16x4*)p = __builtin_convertvector(v, bf16x4);
}

// per-lane validity bits of the keys this lane owns in accumulator layout: bit (4 sb + r) <-> s = 16 sb + 4 g + r
template <typename C> __device__ __forceinline__ void lane_key_bits(const uint8_t* vld, int g, int S, uint32_t& vbits_lo,
                                                                    uint32_t& vbits_hi, uint32_t& ebits_lo, uint32_t& ebits_hi) {
    uint64_t vb = 0, eb = 0;
#pragma unroll
    for (int sb = 0; sb < C::NSB; ++sb) {
        const uint32_t w = *(const uint32_t*)(vld + sb * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if ((w >> (8 * r)) & 0xffu) vb |= (1ull << (sb * 4 + r));
            if (sb * 16 + g * 4 + r < S) eb |= (1ull << (sb * 4 + r));
        }
    }
    vbits_lo = (uint32_t)vb; vbits_hi = (uint32_t)(vb >> 32);
    ebits_lo = (uint32_t)eb; ebits_hi = (uint32_t)(eb >> 32);
}
__device__ __forceinline__ bool bit64(uint32_t lo, uint32_t hi, int i) {
    return i < 32 ? ((lo >> i) & 1u) : ((hi >> (i - 32)) & 1u);
}


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#ifndef ATTN_STORE_AUX
#define ATTN_STORE_AUX 0         // cache policy bits of the attention kernels' output stores (2 = nt, 16 = sc1): timing experiments
#endif
#define OOB 0x80000000u          // byte offset past any descriptor range (< 2 GiB), with room for +offsets without wrapping: loads return 0, stores are dropped

// Hardware-bounds-checked buffer access: rows past T (and the zero-padded channels of D = 16) need no branches, so the
// compiler sees every VMEM op of the loop and can wait with exact vmcnt counts (loads of the NEXT tile stay in flight
// behind this tile's stores instead of draining at vmcnt(0) every iteration).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
template <typename T> __device__ __forceinline__ typename Elem<T>::v8 buf_load8(__amdgpu_buffer_rsrc_t r, uint32_t off);
#ifndef ATTN_LOAD_AUX
#define ATTN_LOAD_AUX 0          // cache policy bits of the streamed Q / dO row loads (2 = nt, 16 = sc1): timing experiments
#endif
template <> __device__ __forceinline__ bf16x8 buf_load8<bf16>(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, ATTN_LOAD_AUX));
}
template <> __device__ __forceinline__ f32x8 buf_load8<float>(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0));
    f32x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return v;
}
template <typename T> __device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, uint32_t off, const f32x4& v);
template <> __device__ __forceinline__ void buf_store4<bf16>(__amdgpu_buffer_rsrc_t r, uint32_t off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4)), r, off, 0, ATTN_STORE_AUX);
}
template <> __device__ __forceinline__ void buf_store4<float>(__amdgpu_buffer_rsrc_t r, uint32_t off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, ATTN_STORE_AUX);
}
template <typename T> __device__ __forceinline__ typename Elem<T>::v4 cvt4(const f32x4& v);
template <> __device__ __forceinline__ f32x4 cvt4<float>(const f32x4& v) { return v; }
template <> __device__ __forceinline__ bf16x4 cvt4<bf16>(const f32x4& v) { return __builtin_convertvector(v, bf16x4); }
template <typename T> __device__ __forceinline__ void buf_store_v4(__amdgpu_buffer_rsrc_t r, uint32_t off, const typename Elem<T>::v4& v);
template <> __device__ __forceinline__ void buf_store_v4<bf16>(__amdgpu_buffer_rsrc_t r, uint32_t off, const bf16x4& v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, off, 0, ATTN_STORE_AUX);
}
template <> __device__ __forceinline__ void buf_store_v4<float>(__amdgpu_buffer_rsrc_t r, uint32_t off, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, ATTN_STORE_AUX);
}
// byte offset of (row t, channel dcol) in a [T, H*D] slab whose descriptor starts at (b, 0, h*D); OOB for padding channels
template <typename T, typename C> __device__ __forceinline__ uint32_t row_off(int t, uint32_t row_bytes, int dcol) {
    uint32_t o = (uint32_t)t * row_bytes + (uint32_t)dcol * (uint32_t)sizeof(T);
    if constexpr (C::D != C::DPAD) o = (dcol < C::D) ? o : OOB;
    return o;
}

// Batched staging of a [S, D] operand slab into an LDS image: ALL of a thread's 16-byte chunks are requested (branch-free buffer
// loads: rows past S and padding channels fall outside the descriptor and read as zero) before the first one is written, so a
// workgroup pays one memory round trip for K, V and the key mask together.  The loop forms above (stage_row_image /
// stage_rowmajor_image) wait for each chunk before requesting the next: with the run-time trip count hipcc keeps one load in
// flight, i.e. 8 + 8 serialized round trips for a 64 x 64 bf16 K and V in a one-wave workgroup.  NT = threads per workgroup.
template <typename T, typename C, int NT> struct ImageStage {
    typedef typename Elem<T>::v8 v8;
    static constexpr int TOTAL = C::SPAD * C::CPR, N = (TOTAL + NT - 1) / NT;
    v8 r[N];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, uint32_t row_bytes) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int i = (int)threadIdx.x + n * NT, s_ = i / C::CPR, c = i % C::CPR;
            r[n] = buf_load8<T>(rs, (i < TOTAL) ? row_off<T, C>(s_, row_bytes, c * 8) : OOB);
        }
    }
    __device__ __forceinline__ void store_row(T* img) const {              // fragment-linear row image (stage_row_image's layout)
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int i = (int)threadIdx.x + n * NT, s_ = i / C::CPR, c = i % C::CPR;
            if (i < TOTAL) *(v8*)(img + rf_idx<C>(s_ >> 4, c >> 2, (s_ & 15) + 16 * (c & 3))) = r[n];
        }
    }
    __device__ __forceinline__ void store_rowmajor(T* img) const {         // row-major padded image (stage_rowmajor_image's layout)
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int i = (int)threadIdx.x + n * NT, s_ = i / C::CPR, c = i % C::CPR;
            if (i < TOTAL) *(v8*)(img + s_ * C::LD + c * 8) = r[n];
        }
    }
};

// bf16 row-per-lane epilogue: a lane owns 4 consecutive channels (8 B) of each 16-channel block.  Swapping the odd
// 16-lane rows of block `a` with the even rows of block `b` (= a + 1) leaves every lane with 8 consecutive channels:
// rows g = 0,2 hold block a's channels 8(g/2)..+7, rows g = 1,3 block b's -- one 16-byte store per lane and 64 contiguous
// bytes per query row per instruction instead of two 8-byte stores with 32-byte segments.
__device__ __forceinline__ void swap16_u32(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void store_pair_bf16(__amdgpu_buffer_rsrc_t r, uint32_t row_byte_off, int blk_a, int g, bf16x4 va, bf16x4 vb) {
    u32x2 a = __builtin_bit_cast(u32x2, va), b = __builtin_bit_cast(u32x2, vb);
    uint32_t a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    swap16_u32(a0, b0);
    swap16_u32(a1, b1);
    const u32x4 v = {a0, a1, b0, b1};
    const uint32_t off = row_byte_off + (uint32_t)((blk_a + (g & 1)) * 32 + (g >> 1) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, ATTN_STORE_AUX);
}

// reductions across the four 16-lane groups that share a query row: v_permlane16_swap / v_permlane32_swap are plain
// VALU ops (no LDS round trip like ds_bpermute): swap(v, v) leaves {lower, upper} halves side by side in the two results.
// NB the two operands must live in DIFFERENT registers (the instruction swaps in place): the empty asm makes the copy
// opaque so the compiler cannot fold it back into one register (same-register swap returns the lower rows twice).
__device__ __forceinline__ void swap16(float v, float& lo, float& hi) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    // inline asm, not __builtin_amdgcn_permlane16_swap: this clang maps BOTH result elements of the builtin to
    // extractvalue 0 (tools/probes/permlane_probe.hip), silently returning the lower rows twice.  s_nop 1 = the two wait
    // states a VALU write needs before a v_permlane read.  After the swap: u = {r0, r0', r2, r2'}, w = {r1, r1', r3, r3'}.
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
    lo = __builtin_bit_cast(float, u);
    hi = __builtin_bit_cast(float, w);
}
__device__ __forceinline__ void swap32(float v, float& lo, float& hi) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
    lo = __builtin_bit_cast(float, u);
    hi = __builtin_bit_cast(float, w);
}
__device__ __forceinline__ float xg_max(float v) {
    float a, b;
    swap16(v, a, b);
    v = fmaxf(a, b);
    swap32(v, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float xg_sum(float v) {
    float a, b;
    swap16(v, a, b);
    v = a + b;
    swap32(v, a, b);
    return a + b;
}
// additive key mask in accumulator layout: 0 for a valid key, -inf otherwise.  Used as the MFMA C-input, so masked
// scores cost no instruction in the loop (exp2(-inf) = 0 drops them from the softmax sum as well).
template <typename C> __device__ __forceinline__ void key_bias(uint32_t vlo, uint32_t vhi, f32x4 (&bias)[C::NSB]) {
#pragma unroll
    for (int sb = 0; sb < C::NSB; ++sb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[sb][r] = bit64(vlo, vhi, sb * 4 + r) ? 0.f : -INFINITY;
}


}  // namespace
