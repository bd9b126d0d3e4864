// Weight gradient on the persistent ping-pong structure of gemm8p.hip (bf16):
//     Out[RB][RA] = scale * sum_m  B[m][rb] * A[m][ra]        (dW[N][K_in] = dy^T x:  A = x [M][K_in], B = dy [M][N])
// replaces: autograd's weight gradient of the trainable nn.Linear layers of the gated cross-attention blocks
//           (reference model/modelling_cross_attention.py:194-199, 273, 352-355 under loss.backward(), run_generation.py:484).
// Both operands are K-MAJOR (the contraction index m is the ROW index of both matrices), so nothing is transposed in memory:
// a unit of the LDS ring is [64 m-rows][128 columns] (16 KiB) exactly as it lies in memory (LDS-DMA, 256-byte rows, the 32-byte
// slot XOR-swizzled by m-row & 7 on the source side) and MFMA fragments come out of it with ds_read_b64_tr_b16 (two reads per
// fragment: k = {4g..4g+3, 16+4g..}; both operands use the same k order, which any contraction allows).
// Same schedule as gemm8p_kernel: 8 waves, wave (wr, wc) owns 128 (rb) x 64 (ra) of a 256 x 256 tile, the two waves of a SIMD
// half a phase apart, one unit per phase (Ba: rb columns 0-63 of each wave's block, Ab-of-this-K-tile: ra columns 32-63,
// Bb: rb 64-127, Aa-of-the-next-K-tile: ra 0-31), issued 6 phases ahead, counted vmcnt, stream continuous across work items.
// Differences in the phase body: the fragment reads are inline-asm transpose reads (hipcc puts vmcnt(0) ahead of the builtin
// form while LDS-DMA writes are in flight) and the fragments of phase p+1 are read inside the MFMA cluster of phase p (T8_ROLL).
// Work item = (output tile, K split): a 2048 x 2048 weight is only 64 tiles, so the M rows are cut into `nsplit` ranges whose
// fp32 partial tiles a fixed-order reduce folds afterwards (deterministic, no atomics).
// Needs (M / nsplit) % 128 == 0 and >= 256; RA % 256 == 0 and RB % 256 == 0 under a K split (unsplit: multiples of 8, ragged last
// tiles are masked at the store).
#include "common.h"
#include "gemm8p.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int T8_UNIT = 16384;
constexpr int T8_LDS = 8 * T8_UNIT;

struct T8Args {
    const bf16* A;       // [M][RA], row stride lda
    const bf16* B;       // [M][RB], row stride ldb
    bf16* Out;           // [RB][RA]  (nsplit == 1)
    float* part;         // [nsplit][RB][RA] fp32 partials (nsplit > 1)
    int RA, RB, Mk;      // Mk = rows contracted per split
    int lda, ldb;
    float scale;
    int accumulate;
    int tiles_a, tiles_b, nsplit, total;
};

#define T8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define T8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define T8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ int t8_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// fragment of 16 columns (32-byte slot `blk` of the unit) x 32 k (k-step ks) of a k-major unit
__device__ __forceinline__ bf16x8 t8_frag(const char* unit, int blk, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int row = ks * 32 + 4 * g + (i >> 2);             // row + 16 has the same (row & 7)
    const char* p = unit + row * 256 + ((blk ^ (row & 7)) << 5) + (i & 3) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * 256));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

__global__ __launch_bounds__(512) void gemm8p_tt_kernel(T8Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = a.Mk >> 6;

    const int G = gridDim.x;
    const int wg = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    // work item -> (A column origin, B column origin, first contracted row); consecutive items of an XCD share the tile
    auto item = [&](int it, int& a0, int& b0, int& split) -> bool {
        const int v = it * G + wg;
        if (v >= a.total) return false;
        split = v % a.nsplit;
        const int tile = v / a.nsplit;
        int ta, tb;
        grouped_tile(tile, a.tiles_b, a.tiles_a, tb, ta);
        a0 = ta * 256;
        b0 = tb * 256;
        return true;
    };
    auto mk_desc = [&](const bf16* base, int col0, int split, int ld, bool valid) {
        const size_t off = (size_t)split * a.Mk * ld + col0;
        // exact end of this split's rows: in a ragged last column tile the columns past the matrix wrap into the next row (finite
        // values whose products land in output rows / columns that are never stored) and read as zero past the last row
        long long rem = valid ? ((long long)a.Mk * ld - col0) * 2 : 0;
        if (rem > 0xffffffffLL) rem = 0xffffffffLL;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (valid ? off : 0)), 0, (int)(unsigned)rem, 0x00020000);
    };

    // ---- staging offsets: unit type 0 Ba, 1 Ab, 2 Bb, 3 Aa; a wave moves pieces `wave` and `wave + 8` (4 m-rows of 256 B each).
    // One lane offset per operand: the b half of a unit type is 128 / 64 bytes further, piece wave + 8 is 32 rows further (both
    // added to the scalar offset) and has the same row & 7, hence the same source chunk.
    int voffB, voffA;
    {
        const int row = wave * 4 + (lane >> 4);
        const int s16 = lane & 15;
        const int c16 = ((((s16 >> 1) ^ (row & 7)) << 1) | (s16 & 1));      // source 16-byte chunk of this LDS position
        const int cl = c16 * 8;                                               // first of its 8 unit columns
        voffB = (row * a.ldb + (cl >> 6) * 128 + (cl & 63)) * 2;
        voffA = (row * a.lda + (cl >> 5) * 64 + (cl & 31)) * 2;
    }
    const int rows32B = 32 * a.ldb * 2, rows32A = 32 * a.lda * 2;
    auto stage = [&](int ty, int slot, __amdgpu_buffer_rsrc_t rs, int soff) __attribute__((always_inline)) {
        lds_void* d0 = (lds_void*)(smem + slot * T8_UNIT + wave * 1024);
        lds_void* d1 = (lds_void*)(smem + slot * T8_UNIT + (wave + 8) * 1024);
        // (the instruction's immediate offset would move the LDS address too: the b halves go through the scalar offset)
        const int vo = (ty & 1) ? voffA : voffB;
        const int so = soff + (ty == 2 ? 128 : ty == 1 ? 64 : 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d0, 16, vo, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, d1, 16, vo, so + ((ty & 1) ? rows32A : rows32B), 0, 0);
    };
    // ---- fragment addresses.  Byte offset of lane (i, g)'s first transpose read of fragment (slot, blk, ks):
    //   slot * 16384 + ks * 8192 + row * 256 + ((blk ^ (row & 7)) << 5) + (i & 3) * 8,   row = 4 g + (i >> 2)
    // The lane part L = row * 256 + (i & 3) * 8 has no bits 5-7, so L + ((blk ^ r7) << 5) = (L | r7 << 5) ^ (blk << 5): ONE lane
    // constant per operand side (with the wave's part of blk folded in: wr << 7 / wc << 6), the fragment index as an XOR with
    // an inline constant at the point of use (v_xor in volatile asm: not hoisted, so not 30 loop-invariant address registers),
    // slot and ks in the instruction's offset field (slots 4-7 through a second base, 64 KiB up).
    int fbase_b, fbase_a;
    {
        const int i = lane & 15, g = lane >> 4;
        const int row = 4 * g + (i >> 2);
        const int L = (row * 256 + (i & 3) * 8) | ((row & 7) << 5);
        const int lds0 = (int)(unsigned)(size_t)(lds_void*)smem;       // 0 (no static LDS in this kernel); a multiple of 1024 by declaration
        fbase_b = (L ^ (wr << 7)) + lds0;
        fbase_a = (L ^ (wc << 6)) + lds0;
    }
    const int fbase_b_hi = fbase_b + 65536, fbase_a_hi = fbase_a + 65536;
    // dst = fragment (SLOT, SUB, KS) of the side whose lane bases are BLO / BHI (SUB: j for B, t for A; all literals).
    // The two transpose reads are INLINE ASM on purpose: behind the builtin, hipcc's waitcnt pass cannot tell the read from the
    // LDS-DMA writes in flight and puts `s_waitcnt vmcnt(0)` in front of every one of them -- the whole 6-unit prefetch drained in
    // every phase (that, not the instruction count, held this kernel at 1.13 PF; the NT kernel's plain 16-byte reads are not
    // affected).  The price: the compiler no longer counts lgkmcnt for these registers, so every phase waits for lgkmcnt(0)
    // right after its barrier, before its first MFMA (the reads were issued a partner-cluster earlier).
#define T8_FRAG(dst, BLO, BHI, SLOT, SUB, KS)                                                                        \
    do {                                                                                                             \
        int ad_;                                                                                                     \
        if ((SUB) == 0) ad_ = ((SLOT) >= 4 ? BHI : BLO);                                                              \
        else asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ad_) : "i"(((SUB) & 3) << 5), "v"((SLOT) >= 4 ? BHI : BLO));   \
        bf16x4 lo_, hi_;                                                                                             \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo_) : "v"(ad_), "i"(((SLOT) & 3) * T8_UNIT + (KS) * 8192)); \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi_) : "v"(ad_), "i"(((SLOT) & 3) * T8_UNIT + (KS) * 8192 + 4096)); \
        dst = bf16x8{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};                                 \
    } while (0)
    // fragments: B side (8 x 16 rb columns per wave, 4 per unit), A side (4 x 16 ra columns per wave, 2 per unit)
    auto rdB = [&](bf16x8 (&f)[2][4], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[0][j] = t8_frag(smem + slot * T8_UNIT, wr * 4 + j, 0, lane);
            f[1][j] = t8_frag(smem + slot * T8_UNIT, wr * 4 + j, 1, lane);
        }
    };
    auto rdA = [&](bf16x8 (&f)[2][2], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f[0][t] = t8_frag(smem + slot * T8_UNIT, wc * 2 + t, 0, lane);
            f[1][t] = t8_frag(smem + slot * T8_UNIT, wc * 2 + t, 1, lane);
        }
    };

    f32x4 acc[4][8];
#define T8_MM(FB, FA, J0, T0, I0, I1)                                                          \
    do {                                                                                       \
        _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_) {                               \
            const int ks_ = i_ >> 3, j_ = (i_ >> 1) & 3, t_ = i_ & 1;                          \
            mma16(acc[(T0) + t_][(J0) + j_], FA[ks_][t_], FB[ks_][j_]);                        \
        }                                                                                      \
    } while (0)

    int a0 = 0, b0 = 0, sp = 0, a1 = 0, b1 = 0, sp1 = 0, it = 0;
    if (!item(0, a0, b0, sp)) return;
    bool have_next = item(1, a1, b1, sp1);
    __amdgpu_buffer_rsrc_t dBc = mk_desc(a.B, b0, sp, a.ldb, true), dAc = mk_desc(a.A, a0, sp, a.lda, true);
    __amdgpu_buffer_rsrc_t dBn = mk_desc(a.B, b1, sp1, a.ldb, have_next), dAn = mk_desc(a.A, a1, sp1, a.lda, have_next);
    const int stepB = 64 * a.ldb * 2, stepA = 64 * a.lda * 2;      // bytes per K tile (scalar offset of the LDS-DMA)

    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] = vzero<f32x4>();
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int ln = t8_lane();
        const int x = ln & 15, g = ln >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t rb = (size_t)(b0 + wr * 128 + j * 16 + x);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ca = a0 + wc * 64 + t * 16 + g * 4;
                const size_t o = rb * a.RA + ca;
                if (a.nsplit > 1) *(f32x4*)(a.part + (size_t)sp * a.RA * a.RB + o) = acc[t][j];
                else if (rb < (size_t)a.RB && ca < a.RA) {            // ragged last tiles (nsplit == 1 only)
                    f32x4 v = acc[t][j] * a.scale;
                    if (a.accumulate) {
                        const bf16x4 ov = *(const bf16x4*)(a.Out + o);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)ov[r];
                    }
                    *(bf16x4*)(a.Out + o) = __builtin_convertvector(v, bf16x4);
                }
            }
        }
        zero_acc();
    };

    // ---- prologue: units -1 .. 5 of the stream  (Aa(0) | Ba(0) Ab(0) Bb(0) Aa(1) | Ba(1) Ab(1))
    stage(3, 7, dAc, 0);
    stage(0, 0, dBc, 0);
    stage(1, 1, dAc, 0);
    stage(2, 2, dBc, 0);
    stage(3, 3, dAc, stepA);
    stage(0, 4, dBc, stepB);
    stage(1, 5, dAc, stepA);
    zero_acc();
    bf16x8 fb[2][4], faA[2][2], faB[2][2];
    T8_VMCNT(10);
    T8_BARRIER();
    rdA(faA, 7);
    rdB(fb, 0);
    T8_LGKM0();
    if (wr) T8_BARRIER();

    // Rolling fragment reloads.  A fragment costs TWO transpose reads, a phase that refills the eight B fragments issues 16 of them,
    // and as a block in front of the MFMAs that does not fit beside the partner wave's 272 MFMA clocks (the NT kernel
    // with half the LDS instructions runs 40 % faster on the same flops).  So no phase reads ahead of its barrier any more: the
    // fragments of phase p+1 are read INSIDE the MFMA cluster of phase p, each into its register right after the last MFMA that
    // uses the old contents (B fragments, and an A buffer that is in use) or spread over the cluster (an A buffer that is free).
    // Unit p+1 has landed by then (the vmcnt + barrier of phase p already guaranteed it), its slot is re-staged three phases later.
#define T8_M1(FB, FA, J0, T0, i) mma16(acc[(T0) + ((i) & 1)][(J0) + (((i) >> 1) & 3)], FA[(i) >> 3][(i) & 1], FB[(i) >> 3][((i) >> 1) & 3])
#define T8_FRB(S, q) T8_FRAG(fb[(q) >> 2][(q) & 3], fbase_b, fbase_b_hi, S, (q) & 3, (q) >> 2)
#define T8_FRA(F, S, ks, t) T8_FRAG(F[ks][t], fbase_a, fbase_a_hi, S, t, ks)
    // hooks: what is read after MFMA pair q (q = 0..7; pair 7's second MFMA runs after the hand-over barrier)
#define T8_HK_B(S, q) T8_FRB(S, q)                                                      /* refill fb, pair by pair */
#define T8_HK_AFREE(F, S, q) do { if (!((q) & 1)) T8_FRA(F, S, (q) >> 2, ((q) >> 1) & 1); } while (0)   /* a free A buffer: after pairs 0, 2, 4, 6 */
#define T8_HK_AUSED(F, S, q) do { if ((q) == 3) { T8_FRA(F, S, 0, 0); T8_FRA(F, S, 0, 1); } if ((q) == 7) { T8_FRA(F, S, 1, 0); T8_FRA(F, S, 1, 1); } } while (0)
#define T8_PHASE2(TY, SLOT, DK, FB, FA, J0, T0, HOOK)                                            \
    do {                                                                                         \
        {                                                                                        \
            const int kk_ = kt + (DK);                                                           \
            const bool nx_ = kk_ >= nk;                                                          \
            const int ki_ = nx_ ? kk_ - nk : kk_;                                                \
            if ((TY) & 1) stage((TY), (SLOT), nx_ ? dAn : dAc, ki_ * stepA);                     \
            else stage((TY), (SLOT), nx_ ? dBn : dBc, ki_ * stepB);                              \
        }                                                                                        \
        T8_VMCNT(10);                                                                            \
        T8_BARRIER();                                                                            \
        T8_LGKM0();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_setprio(1);                                                           \
        T8_M1(FB, FA, J0, T0, 0); T8_M1(FB, FA, J0, T0, 1); HOOK(0); __builtin_amdgcn_sched_barrier(0);     \
        T8_M1(FB, FA, J0, T0, 2); T8_M1(FB, FA, J0, T0, 3); HOOK(1); __builtin_amdgcn_sched_barrier(0);     \
        T8_M1(FB, FA, J0, T0, 4); T8_M1(FB, FA, J0, T0, 5); HOOK(2); __builtin_amdgcn_sched_barrier(0);     \
        T8_M1(FB, FA, J0, T0, 6); T8_M1(FB, FA, J0, T0, 7); HOOK(3); __builtin_amdgcn_sched_barrier(0);     \
        T8_M1(FB, FA, J0, T0, 8); T8_M1(FB, FA, J0, T0, 9); HOOK(4); __builtin_amdgcn_sched_barrier(0);     \
        T8_M1(FB, FA, J0, T0, 10); T8_M1(FB, FA, J0, T0, 11); HOOK(5); __builtin_amdgcn_sched_barrier(0);   \
        T8_M1(FB, FA, J0, T0, 12); T8_M1(FB, FA, J0, T0, 13); HOOK(6); __builtin_amdgcn_sched_barrier(0);   \
        T8_M1(FB, FA, J0, T0, 14);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        T8_BARRIER();                                                                            \
        T8_M1(FB, FA, J0, T0, 15);                                                               \
        HOOK(7);                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
#define T8_H1(q) T8_HK_AFREE(faB, 1, q)
#define T8_H2(q) T8_HK_B(2, q)
#define T8_H3(q) T8_HK_AUSED(faB, 3, q)
#define T8_H4(q) T8_HK_B(4, q)
#define T8_H5(q) T8_HK_AFREE(faA, 5, q)
#define T8_H6(q) T8_HK_B(6, q)
#define T8_H7(q) T8_HK_AUSED(faA, 7, q)
#define T8_H8(q) T8_HK_B(0, q)

    for (;;) {
        for (int kt = 0; kt < nk; kt += 2) {
            T8_PHASE2(2, 6, 1, fb, faA, 0, 0, T8_H1);      // Ba x Aa      | reads Ab   -> faB
            T8_PHASE2(3, 7, 2, fb, faB, 0, 2, T8_H2);      // Ba x Ab      | reads Bb   -> fb
            T8_PHASE2(0, 0, 2, fb, faB, 4, 2, T8_H3);      // Bb x Ab      | reads Aa'  -> faB
            T8_PHASE2(1, 1, 2, fb, faA, 4, 0, T8_H4);      // Bb x Aa      | reads Ba'  -> fb
            T8_PHASE2(2, 2, 2, fb, faB, 0, 0, T8_H5);      // Ba' x Aa'    | reads Ab'  -> faA
            T8_PHASE2(3, 3, 3, fb, faA, 0, 2, T8_H6);      // Ba' x Ab'    | reads Bb'  -> fb
            T8_PHASE2(0, 4, 3, fb, faA, 4, 2, T8_H7);      // Bb' x Ab'    | reads Aa'' -> faA
            T8_PHASE2(1, 5, 3, fb, faB, 4, 0, T8_H8);      // Bb' x Aa'    | reads Ba'' -> fb
        }
        epilogue();
        if (!have_next) break;
        ++it;
        a0 = a1;
        b0 = b1;
        sp = sp1;
        dAc = dAn;
        dBc = dBn;
        have_next = item(it + 1, a1, b1, sp1);
        dBn = mk_desc(a.B, b1, sp1, a.ldb, have_next);
        dAn = mk_desc(a.A, a1, sp1, a.lda, have_next);
    }
    T8_VMCNT(0);
    if (!wr) T8_BARRIER();
}

}  // namespace

// number of K splits (0 = shape not eligible): enough (tile, split) work items to fill the chip, whole K-tile pairs per split.
// Few-tile outputs (a 2048 x 2048 weight: 64 tiles; a rank-padded LoRA factor: 8) take up to 16 splits and run with as few as
// 64 work items -- still far ahead of the non-persistent 128 x 128 kernel, which has no K split at all.
int gemm8p_tt_splits(int RA, int RB, int M) {
    if (M % 128) return 0;
    if (RA % 256 || RB % 256)                        // ragged outputs (lm_head: 50272 rows): unsplit only, when the tiles fill the chip
        return (RA % 8 == 0 && RB % 8 == 0 && M >= 256 && cdiv(RA, 256) * cdiv(RB, 256) >= 192) ? 1 : 0;
    // every split count that cuts the M rows into equal runs of whole 128-row steps, priced in 128-row steps (~1.7 us of a CU):
    // rounds of work items x (steps per item + ~3 of per-item fixed cost), and for a split the fp32 partial tiles -- 3 more steps per
    // item to write them plus 0.06 per partial tile for the fold (256 KiB written and read back at ~5 TB/s, chip-wide).  A 768 x 768
    // gradient over 40960 rows (9 tiles) takes 20 splits (180 items of 16 steps) instead of 16 (144 of 20), a 3072 x 768 one 5
    // (180 x 64) instead of 8 (288 items = two rounds of 40); 64 tiles stay at 4 splits, outputs of >= 256 tiles unsplit.
    const int tiles = (RA / 256) * (RB / 256), units = M / 128, G = gemm8p_num_cu();
    int best = 0, best_cost = 0;
    for (int s = 1; s <= 32; ++s) {
        if (units % s || M / s < 256) continue;
        const int cost = cdiv(tiles * s, G) * (units / s + 3) + (s > 1 ? 3 + tiles * s * 6 / 100 : 0);
        if (!best || cost < best_cost) { best = s; best_cost = cost; }
    }
    return (best && tiles * best >= 64) ? best : 0;
}

int launch_gemm8p_tt(const bf16* A, int lda, const bf16* B, int ldb, bf16* Out, float* part, int RA, int RB, int M, int nsplit, float scale,
                     int accumulate, hipStream_t st) {
    if (nsplit < 1 || ((RA % 256 || RB % 256) && (nsplit > 1 || RA % 8 || RB % 8)) || M % (128 * nsplit) || M / nsplit < 256 || (nsplit > 1 && !part))
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm8p_tt: shape RA=%d RB=%d M=%d nsplit=%d not supported", RA, RB, M, nsplit);
    if ((long long)M * lda * 2 >= 0xffffffffLL || (long long)M * ldb * 2 >= 0xffffffffLL)
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm8p_tt: operand larger than 4 GiB");
    T8Args a;
    a.A = A; a.B = B; a.Out = Out; a.part = part; a.RA = RA; a.RB = RB; a.Mk = M / nsplit; a.lda = lda; a.ldb = ldb; a.scale = scale;
    a.accumulate = accumulate; a.tiles_a = cdiv(RA, 256); a.tiles_b = cdiv(RB, 256); a.nsplit = nsplit; a.total = a.tiles_a * a.tiles_b * nsplit;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "gemm8p_tt: hipGetDeviceProperties failed");
        n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        hipError_t e = hipFuncSetAttribute((const void*)gemm8p_tt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS);
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int grid = a.total < n_cu ? a.total : n_cu;
    hipLaunchKernelGGL(gemm8p_tt_kernel, dim3(grid), dim3(512), T8_LDS, st, a);
    MMGL_CHECK_LAUNCH("gemm8p_tt");
    return MMGL_OK;
}
