// Persistent 256x256 NT GEMM for gfx950 (bf16) on FOUR waves per CU -- one per SIMD, the whole 512-register budget each:
//   Y[M,N] = epi( X[M,K] . W[N,K]^T )
// replaces: the same nn.Linear calls as gemm8p.hip (reference model/modelling_cross_attention.py:194-199, :273, :352-355, :826 and their
//           dgrads); it exists to take the TILE BOUNDARY out of the critical path (DESIGN 9.6c iv-a): the eight-wave ping-pong kernel has
//           no registers left to hold a finished tile, so its output stores (128 KiB per CU) sit in the CU's in-order memory pipe ahead of
//           the next tile's operand loads -- 18 % of a K = 2048 tile.  Here a wave owns a 128x128 block (256 fp32 accumulators in the
//           AGPR half of the file), converts a finished tile to 128 packed bf16 registers and stores them ONE INSTRUCTION AT A TIME under
//           the next tile's MFMAs.
//
// Structure:
//   * wave (wr, wc) = (wave >> 1, wave & 1) owns rows [128 wr, +128) x columns [128 wc, +128) of the tile as 4 x 4 blocks of
//     v_mfma_f32_32x32x16_bf16.  W is the MFMA A operand with its fragment rows permuted in the staging addresses
//     (fragment row 8 q + 4 h + r holds n = 16 h + 4 q + r) so that lane (c, h) ends with 16 CONSECUTIVE output columns of row c per
//     block; one v_permlane16_swap per register pair then gives four lanes of a row 64 contiguous bytes per store instruction
//     (16 rows x 64 B: the store shape of gemm8p).
//   * operands stream global -> LDS by LDS-DMA in units of 128 rows x 128 B (one K step of 64) in the order X0 X1 W0 W1 per K step
//     through a ring of NINE unit slots; a wave reads unit X[wr] and W[wc] of a step: four substeps of 16 MFMAs, the fragments of
//     substep t + 1 read (8 ds_read_b128) between the MFMAs of substep t into the other of two fragment register sets.
//   * ONE barrier per K step (after substep 2: everybody has read the step's last fragments, everybody's pieces of the next step have
//     landed).  The unit that is 6 substeps ahead is requested in each substep; the four waves take turns (wave w issues in MFMA
//     gaps w, w + 4, w + 8, w + 12) so that the CU's one global -> LDS path (64 B/clk: 16 clocks per 1 KiB piece) never sees two
//     requests in one 32-clock MFMA gap -- a wave that waits for that path cannot issue its next MFMA.
//   * the stream of units runs across output tiles (no pipeline fill per tile); the first E = 32 / S K steps of a tile each carry S
//     store instructions of the PREVIOUS tile (compile-time register indices: those steps are unrolled), so needs K >= 64 E.
#include "common.h"
#include "gemm8p.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned g4_u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int G4_UNIT = 16384;
constexpr int G4_NSLOT = 9;
constexpr int G4_RING = G4_NSLOT * G4_UNIT;
constexpr int G4_BIAS = G4_RING;                    // two 1 KiB bias lines per wave (tile parity): an LDS-DMA instruction deposits 1 KiB
constexpr int G4_LDS_ALLOC = G4_BIAS + 8192 + 64;
constexpr int G4_LEAD = 6;                          // units requested ahead of the substep that runs

#ifndef G4_STORE_AUX
#define G4_STORE_AUX 18                             // nt + sc1 (gemm8p.hip: the output streams past the L2 that holds the operand slices)
#endif
#ifndef G4_TRACE
#define G4_TRACE 0
#endif

struct G4Args {
    const bf16* X;
    const bf16* W;
    bf16* Y;
    const bf16* bias;
    int M, N, K;
    int ldx, ldw, ldy;
    float scale;
    int tiles_m, tiles_n, total;
    long long* trace;
    int trace_wg;
};

#define G4_BARRIER() asm volatile("s_barrier" ::: "memory")
#define G4_WAIT(vm) asm volatile("s_waitcnt vmcnt(" #vm ") lgkmcnt(0)" ::: "memory")
#define G4_VMCNT(vm) asm volatile("s_waitcnt vmcnt(" #vm ")" ::: "memory")

__device__ __forceinline__ f32x16 g4_mma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// u = {u.r0, w.r0, u.r2, w.r2}, w = {u.r1, w.r1, u.r3, w.r3} (rows of 16 lanes); inline asm: this clang maps both results of the
// builtin to element 0 (tools/probes/permlane_probe.hip); s_nop 1 = the wait states between a VALU write and a v_permlane read
__device__ __forceinline__ void g4_swap16(unsigned& u, unsigned& w) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
}
__device__ __forceinline__ unsigned g4_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int I> using g4_c = std::integral_constant<int, I>;

// S = store instructions of the previous tile per K step (4: K >= 512; 8: K >= 256)
// The kernel body exists once per wave of the workgroup (WV is a compile-time constant): the four waves run the same schedule
// except for the MFMA gaps their LDS-DMA requests sit in.  Measured (profiles/r5_gemm4w_*): a request costs the issuing wave ~56
// clocks when all four waves issue in the same gap (the CU's one global -> LDS path serves them one after the other), a conditional
// branch around a request ~50 when taken, a request under EXEC = 0 as much as a live one -- so whose turn it is has to be known
// at compile time.
template <int ACT, int S, int WV> __device__ __forceinline__ void gemm4w_body(const G4Args& a, char* smem) {
    constexpr int E = 32 / S;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int wave = WV;
    constexpr int wr = wave >> 1, wc = wave & 1;
    const int nk = a.K >> 6;

    // ---- static persistent schedule (gemm8p's: per round of gridDim tiles XCD j takes 32 consecutive virtual ids)
    const int G = gridDim.x;
    const int wg = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    auto item_origin = [&](int v, int& m0, int& n0) -> bool {
        if (v < 0 || v >= a.total) return false;
        int tm, tn;
        grouped_tile(v, a.tiles_m, a.tiles_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
        return true;
    };
    auto mk_desc = [&](const bf16* base, int row0, int rows, int ld, bool valid) {
        long long rem = valid ? (long long)(rows - row0) * ld * 2 : 0;
        if (rem > 0xffffffffLL) rem = 0xffffffffLL;
        if (rem < 0) rem = 0;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (valid ? (size_t)row0 * ld : 0)), 0, (int)(unsigned)rem, 0x00020000);
    };

    // ---- staging: piece p = wave + 4 i of a unit = LDS rows 8 p .. 8 p + 7; 16-byte position (lane & 7) of LDS row lr holds the
    // k slot (lane & 7) ^ ((lr >> 1) & 7); X: LDS row = tile row; W: LDS row 32 b + 8 q + 4 h + r = tile row 32 b + 16 h + 4 q + r
    // (the second unit of a step, 128 rows on, has lane offsets of its own: the descriptor's range check -- rows past M / N read as
    // zero -- covers the lane offset only, not the scalar one)
    // Piece p: row 8 p + r8 (r8 = lane >> 3), k slot (lane & 7) ^ ((4 (p & 1) + (r8 >> 1)) & 7): the lane part depends on the PARITY of p
    // only, the rest of p is a scalar row offset added per request (asm volatile: else every sum is hoisted into a register).
    // X: row offset 8 p;  W: tile row 32 (p >> 2) + 4 (p & 3) + [16 ((r8 >> 2) & 1) + (r8 & 3)].
    int voffXp[2], voffWp[2];
    {
        const int r8 = lane >> 3;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int cs = (lane & 7) ^ ((4 * par + (r8 >> 1)) & 7);
            voffXp[par] = r8 * a.ldx * 2 + cs * 16;
            voffWp[par] = (16 * ((r8 >> 2) & 1) + (r8 & 3)) * a.ldw * 2 + cs * 16;
        }
    }
    const int ldx2 = a.ldx * 2, ldw2 = a.ldw * 2;
    const int voffXme = ((wave & 1) ? voffXp[1] : voffXp[0]) + 8 * wave * ldx2, voffWme = ((wave & 1) ? voffWp[1] : voffWp[0]) + 4 * wave * ldw2;
    // ---- fragment addressing: lane (c, h) reads row c of a 32-row block, 16-byte position (2 ks + h) ^ ((c >> 1) & 7)
    // = lb ^ (ks << 5) with one lane constant
    int lb;
    {
        const int c = lane & 31, h = lane >> 5, key = (c >> 1) & 7;
        lb = c * 128 + (((h ^ key) & 1) << 4) + ((key >> 1) << 5);
    }

    f32x16 acc[4][4];
    bf16x8 fw[2][4], fx[2][4];
    g4_u32x4 pk[32];                                  // the previous tile, packed: store q = (i * 4 + j) * 2 + half
#pragma unroll
    for (int q = 0; q < 32; ++q) asm volatile("" : "=v"(pk[q]));      // (undefined until the first tile is converted: its stores meet an empty descriptor)
    int voffb[4] = {0, 0, 0, 0};                      // ... and its lane offsets by column block (0x80000000: past N)
    __amdgpu_buffer_rsrc_t dYp = __builtin_amdgcn_make_buffer_rsrc((void*)a.Y, 0, 0, 0x00020000);     // empty: nothing pending
    const int rstep16 = 16 * a.ldy * 2;

#if G4_TRACE
    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (a.trace && (int)blockIdx.x == a.trace_wg && tr_n < 96) {
            const long long t = __builtin_readcyclecounter();
            if (lane == 0) a.trace[wave * 96 + tr_n] = t;
            ++tr_n;
        }
    };
#define G4_STAMP() stamp()
#else
#define G4_STAMP() (void)0
#endif

    int m0 = 0, n0 = 0, m1 = 0, n1 = 0;
    int it = 0;
    if (!item_origin(wg, m0, n0)) return;
    bool have_next = item_origin(G + wg, m1, n1);
    __amdgpu_buffer_rsrc_t dXc = mk_desc(a.X, m0, a.M, a.ldx, true), dWc = mk_desc(a.W, n0, a.N, a.ldw, true);
    __amdgpu_buffer_rsrc_t dXn = mk_desc(a.X, m1, a.M, a.ldx, have_next), dWn = mk_desc(a.W, n1, a.N, a.ldw, have_next);
    const __amdgpu_buffer_rsrc_t dBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 2 : 0, 0x00020000);
    auto fetch_bias = [&](int nt0, int par) __attribute__((always_inline)) {
        const unsigned off = lane < 16 ? (unsigned)((nt0 + wc * 128 + lane * 8) * 2) : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dBias, (lds_void*)(smem + G4_BIAS + (par & 1) * 4096 + wave * 1024), 16, off, 0, 0, 0);
    };

    // lane offset of a request = lane part + (row count of the piece) * (row pitch), computed at the request in one asm volatile
    // statement: left to the compiler every one of these loop-invariant products is hoisted into an SGPR and spilled
#define G4_VO(vo_, rows_, pitch_, lanepart_)                                                                                  \
    do {                                                                                                                      \
        int t_;                                                                                                               \
        asm volatile("s_mul_i32 %1, %2, %3\n\tv_add_u32 %0, %1, %4" : "=v"(vo_), "=&s"(t_) : "s"(pitch_), "i"(rows_), "v"(lanepart_)); \
    } while (0)
    // the four pieces of a unit this wave moves when the waves share it: wave + 4 i (piece parity = wave parity; the wave's own
    // rows -- 8 wave (X), 4 wave (W) -- are part of the lane offset)
    auto stage_piece = [&](auto isw_c, auto jj_c, int slot, auto i_c, __amdgpu_buffer_rsrc_t rs, int kbyte) __attribute__((always_inline)) {
        constexpr bool isW = decltype(isw_c)::value != 0;
        constexpr int jj = decltype(jj_c)::value, i = decltype(i_c)::value;
        int vo;
        if constexpr (isW) G4_VO(vo, 128 * jj + 32 * i, ldw2, voffWme);
        else G4_VO(vo, 128 * jj + 32 * i, ldx2, voffXme);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * G4_UNIT + (wave + 4 * i) * 1024), 16, vo, kbyte, 0, 0);
    };
    // ---- prologue: units 0 .. 5 = X0 X1 W0 W1 of K step 0, X0 X1 of K step 1
    {
        auto unit = [&](auto w_c, auto jj_c, int slot, __amdgpu_buffer_rsrc_t rs, int kb) __attribute__((always_inline)) {
            stage_piece(w_c, jj_c, slot, g4_c<0>(), rs, kb);
            stage_piece(w_c, jj_c, slot, g4_c<1>(), rs, kb);
            stage_piece(w_c, jj_c, slot, g4_c<2>(), rs, kb);
            stage_piece(w_c, jj_c, slot, g4_c<3>(), rs, kb);
        };
        unit(g4_c<0>(), g4_c<0>(), 0, dXc, 0);
        unit(g4_c<0>(), g4_c<1>(), 1, dXc, 0);
        unit(g4_c<1>(), g4_c<0>(), 2, dWc, 0);
        unit(g4_c<1>(), g4_c<1>(), 3, dWc, 0);
        unit(g4_c<0>(), g4_c<0>(), 4, dXc, 128);
        unit(g4_c<0>(), g4_c<1>(), 5, dXc, 128);
    }
    fetch_bias(n0, 0);
    int sq = 0;                                       // (4 * K steps done) mod 9: slot of the running step's X0
    G4_VMCNT(9);                                      // units 0 .. 3 have landed (units 4, 5 and the bias line may be in flight)
    G4_BARRIER();
    {
        const int vX = lb + wr * G4_UNIT, vW = lb + (2 + wc) * G4_UNIT;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            fw[0][b] = *(const bf16x8*)(smem + vW + b * 4096);
            fx[0][b] = *(const bf16x8*)(smem + vX + b * 4096);
        }
    }

    // One substep = 16 MFMAs of K step `st`.  KS: substep; SIDX: index of this K step among the tile's first E (it carries the
    // stores SIDX * S .. + S - 1 of the previous tile), or -1.
    int st = 0;
    auto substep = [&](auto ks_c, auto sidx_c) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_c)::value, SIDX = decltype(sidx_c)::value;
        constexpr int cur = KS & 1, nxt = cur ^ 1;
        // the unit requested now: 4 st + KS + 6 = W0 / W1 of K step st + 1 (KS = 0, 1), X0 / X1 of K step st + 2 (KS = 2, 3)
        constexpr int isW = KS < 2 ? 1 : 0;
        constexpr int jj = KS & 1;
        const int stq = st + (KS < 2 ? 1 : 2);
        const bool nx = stq >= nk;
        const int kb = (nx ? stq - nk : stq) * 128;
        const __amdgpu_buffer_rsrc_t rs = isW ? (nx ? dWn : dWc) : (nx ? dXn : dXc);
        int slot = sq + KS + G4_LEAD;
        slot = slot >= G4_NSLOT ? slot - G4_NSLOT : slot;
        // the fragments read now: substep KS + 1 of this K step, or substep 0 of the next one
        int sb = KS == 3 ? sq + 4 : sq;
        sb = sb >= G4_NSLOT ? sb - G4_NSLOT : sb;
        int sx = sb + wr, sw = sb + 2 + wc;
        sx = sx >= G4_NSLOT ? sx - G4_NSLOT : sx;
        sw = sw >= G4_NSLOT ? sw - G4_NSLOT : sw;
        int vX, vW;                                   // (asm volatile: computed here, not kept across the loop)
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(vX) : "s"((sx * G4_UNIT) | (((KS + 1) & 3) << 5)), "v"(lb));
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(vW) : "s"((sw * G4_UNIT) | (((KS + 1) & 3) << 5)), "v"(lb));
        __builtin_amdgcn_sched_barrier(0);
        {
            // this wave's requests: gaps wave, wave + 4, wave + 8, wave + 12 (the timing experiments of profiles/r5_gemm4w_ablation.txt -- every
            // wave in the same gaps, EXEC masks, branches, back-to-back requests, one wave per substep, and the wrong-result ablations --
            // are in the history of this file: commit "gemm4w: four-wave persistent GEMM ...", not in the product source)
            auto dma = [&](auto g_c) __attribute__((always_inline)) {
                constexpr int g = decltype(g_c)::value;
                if constexpr ((g & 3) == wave) stage_piece(g4_c<isW>(), g4_c<jj>(), slot, g4_c<(g >> 2)>(), rs, kb);
            };
            auto gap = [&](auto g_c) __attribute__((always_inline)) {
                constexpr int g = decltype(g_c)::value;
                constexpr int i = g >> 2, j = (i & 1) ? 3 - (g & 3) : (g & 3);
                constexpr bool bar = KS == 2 && g == 15;
                if constexpr (bar) {
                    // the K step's barrier, ahead of the substep's last MFMA: every wave has read the step's last fragments (its slots
                    // may be overwritten) and the pieces of the next step's four units have landed.  Younger than those pieces: this
                    // substep's requests (4 per wave, or 16 of one wave) and its S stores of the previous tile.
                    dma(g_c);
                    constexpr int young = 4 + (SIDX >= 0 ? S : 0);
                    if constexpr (young == 4) G4_WAIT(4);
                    if constexpr (young == 8) G4_WAIT(8);
                    if constexpr (young == 12) G4_WAIT(12);
                    G4_BARRIER();
                }
                if constexpr (SIDX == 0 && KS == 0) acc[i][j] = g4_mma(fw[cur][i], fx[cur][j], vzero<f32x16>());
                else acc[i][j] = g4_mma(fw[cur][i], fx[cur][j], acc[i][j]);
                // fragments of the next substep, in the order its MFMAs want them: W0 X0 X1 X2 X3 W1 W2 W3
                if constexpr (g == 0) fw[nxt][0] = *(const bf16x8*)(smem + vW);
                if constexpr (g >= 1 && g <= 4) fx[nxt][g - 1] = *(const bf16x8*)(smem + vX + (g - 1) * 4096);
                if constexpr (g >= 5 && g <= 7) fw[nxt][g - 4] = *(const bf16x8*)(smem + vW + (g - 4) * 4096);
                if constexpr (!bar) dma(g_c);
                if constexpr (SIDX >= 0 && KS == 2) {
                    constexpr int every = 16 / S;
                    if constexpr ((g % every) == every / 2 && g / every < S) {
                        constexpr int q = SIDX * S + g / every;
                        constexpr int qi = q >> 3, qr = q & 7;          // column block, (row block, half)
                        int vo;                                         // (asm volatile: else all 32 sums are hoisted out of the K loop)
                        G4_VO(vo, qr, rstep16, voffb[qi]);
                        __builtin_amdgcn_raw_buffer_store_b128(pk[q], dYp, vo, 0, G4_STORE_AUX);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            gap(g4_c<0>()); gap(g4_c<1>()); gap(g4_c<2>()); gap(g4_c<3>()); gap(g4_c<4>()); gap(g4_c<5>()); gap(g4_c<6>()); gap(g4_c<7>());
            gap(g4_c<8>()); gap(g4_c<9>()); gap(g4_c<10>()); gap(g4_c<11>()); gap(g4_c<12>()); gap(g4_c<13>()); gap(g4_c<14>()); gap(g4_c<15>());
        }
    };
    auto kstep = [&](auto sidx_c) __attribute__((always_inline)) {
        substep(g4_c<0>(), sidx_c);
        substep(g4_c<1>(), sidx_c);
        substep(g4_c<2>(), sidx_c);
        substep(g4_c<3>(), sidx_c);
        sq = sq + 4 >= G4_NSLOT ? sq + 4 - G4_NSLOT : sq + 4;
        ++st;
    };

    for (;;) {
        G4_STAMP();
        st = 0;
        kstep(g4_c<0>());
        kstep(g4_c<1>());
        kstep(g4_c<2>());
        kstep(g4_c<3>());
        if constexpr (E == 8) {
            kstep(g4_c<4>());
            kstep(g4_c<5>());
            kstep(g4_c<6>());
            kstep(g4_c<7>());
        }
        G4_STAMP();
        while (st < nk) kstep(g4_c<-1>());
        G4_STAMP();
        // ---- the finished tile -> pk (bias, scale, activation, bf16, 64-byte row segments per four lanes)
        {
            const int h = lane >> 5;
            const int colw = n0 + wc * 128 + 8 * (lane >> 4);
            const int lanepart = ((lane & 15) * a.ldy + 8 * (lane >> 4)) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                voffb[i] = colw + 32 * i < a.N ? lanepart + 64 * i : (int)0x80000000;
                // (no bias: the line holds zeros -- an empty descriptor deposits zeros)
                const char* bl = smem + G4_BIAS + (it & 1) * 4096 + wave * 1024 + (32 * i + 16 * h) * 2;
                const g4_u32x4 bb0 = *(const g4_u32x4*)bl, bb1 = *(const g4_u32x4*)(bl + 16);     // packed: converted pair by pair below
                auto blo = [](unsigned w) { return __builtin_bit_cast(float, w << 16); };
                auto bhi = [](unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); };
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned P[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        P[e] = g4_pk((acc[i][j][2 * e] + blo(bb0[e])) * a.scale, (acc[i][j][2 * e + 1] + bhi(bb0[e])) * a.scale);
                        P[4 + e] = g4_pk((acc[i][j][8 + 2 * e] + blo(bb1[e])) * a.scale, (acc[i][j][8 + 2 * e + 1] + bhi(bb1[e])) * a.scale);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) g4_swap16(P[e], P[4 + e]);
                    pk[(i * 4 + j) * 2] = g4_u32x4{P[0], P[1], P[2], P[3]};
                    pk[(i * 4 + j) * 2 + 1] = g4_u32x4{P[4], P[5], P[6], P[7]};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // rows [m0 + 128 wr, +128) x columns [n0 + 128 wc, ..): rows past M fall outside the descriptor
            long long rem = (long long)(a.M - m0 - wr * 128) * a.ldy * 2 - (long long)(n0 + wc * 128) * 2;
            const long long cap = (long long)128 * a.ldy * 2;
            rem = rem > cap ? cap : rem;
            const bool ok = rem > 0;
            dYp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.Y + (ok ? (size_t)(m0 + wr * 128) * a.ldy + n0 + wc * 128 : 0)), 0, ok ? (int)rem : 0,
                                                    0x00020000);
        }
        G4_STAMP();
        if (!have_next) break;
        ++it;
        m0 = m1;
        n0 = n1;
        dXc = dXn;
        dWc = dWn;
        have_next = item_origin((it + 1) * G + wg, m1, n1);
        dXn = mk_desc(a.X, m1, a.M, a.ldx, have_next);
        dWn = mk_desc(a.W, n1, a.N, a.ldw, have_next);
        fetch_bias(n0, it);
    }
    // ---- the last tile's stores
#pragma unroll
    for (int q = 0; q < 32; ++q)
        __builtin_amdgcn_raw_buffer_store_b128(pk[q], dYp, voffb[q >> 3] + (q & 7) * rstep16, 0, G4_STORE_AUX);
    G4_VMCNT(0);                                      // no LDS-DMA may outlive the workgroup
}

template <int ACT, int S> __global__ __launch_bounds__(256) void gemm4w_kernel(G4Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave == 0) gemm4w_body<ACT, S, 0>(a, smem);
    else if (wave == 1) gemm4w_body<ACT, S, 1>(a, smem);
    else if (wave == 2) gemm4w_body<ACT, S, 2>(a, smem);
    else gemm4w_body<ACT, S, 3>(a, smem);
}

}  // namespace

bool gemm4w_supported(int M, int N, int K, int ldx, int ldw, int ldy) {
    return gemm8p_supported(M, N, K, ldx, ldw, ldy) && K % 64 == 0 && K >= 512 && (long long)256 * ldy * 2 < 0x7fffffffLL;
}

int launch_gemm4w(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, int M, int N, int K, float scale,
                  hipStream_t st) {
    if (!gemm4w_supported(M, N, K, ldx, ldw, ldy)) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm4w: shape M=%d N=%d K=%d not supported", M, N, K);
    G4Args a;
    a.X = X; a.W = W; a.Y = Y; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.scale = scale;
    a.tiles_m = cdiv(M, 256); a.tiles_n = cdiv(N, 256); a.total = a.tiles_m * a.tiles_n;
    a.trace = nullptr;
    a.trace_wg = 0;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm4w_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS_ALLOC);
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
#if G4_TRACE
    if (const char* e = getenv("MMGL_G4_TRACE")) a.trace = (long long*)strtoull(e, nullptr, 0);
    if (const char* e = getenv("MMGL_G4_TRACE_WG")) a.trace_wg = atoi(e);
#endif
    const int n_cu = gemm8p_num_cu();
    const int grid = a.total < n_cu ? a.total : n_cu;
    hipLaunchKernelGGL((gemm4w_kernel<0, 4>), dim3(grid), dim3(256), G4_LDS_ALLOC, st, a);
    MMGL_CHECK_LAUNCH("gemm4w");
    return MMGL_OK;
}
