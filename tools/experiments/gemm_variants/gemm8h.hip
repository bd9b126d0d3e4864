// Persistent ping-pong NT GEMM for gfx950 (bf16) with the two wave groups of a CU on SEPARATE 128x256 half-tiles (DESIGN 9.6c iv-b):
//   Y[M,N] = (X[M,K] . W[N,K]^T + bias) * scale
// replaces: the same nn.Linear calls as gemm8p.hip (reference model/modelling_cross_attention.py:194-199, :273, :352-355, :826 and their
//           dgrads) for plain-epilogue shapes; it exists to run one group's epilogue BESIDE the other group's MFMAs instead of stopping
//           the matrix pipes of the whole CU at every tile boundary (18 % of a K = 2048 tile in gemm8p, profiles/r4_gemm8p_tile_boundary_anatomy.txt).
//
// What is gemm8p's and stays: 8 waves, wave (wr, wc) = (group, 64-column quarter), a 128 (m) x 64 (n) block of fp32 accumulators per
// wave, the two waves of a SIMD half a phase apart, units of 128 LDS rows x 128 B streamed by LDS-DMA through an 8-slot ring (Xa, Wb, Xb,
// Wa-of-the-next-K-tile per K tile of 64), counted vmcnt, the swizzle, the W row permutation, the store shape.
// What is new:
//   * a CU owns a run of consecutive 128-row HALF-tiles of ONE tile column.  Group A (waves 0-3) takes the even ones, group B the odd
//     ones; an X unit carries 64 rows of A's half-tile and 64 rows of B's (the row offset of each rides in the scalar offset of the
//     request), every W unit serves both groups.
//   * the operand stream never stops and its K pointer WRAPS: a group starts a half-tile at whatever K tile the stream is at, computes
//     K / 64 K tiles with wrap-around (any rotation of the K range is the same sum) and then spends ONE cycle (8 phases = 2 K tiles of
//     the stream) on its epilogue -- 16 sub-steps of one store each, two per phase, in the slots where its MFMA cluster and its
//     fragment reads would be -- while the other group computes.  It keeps taking part in both barriers of every phase and keeps
//     issuing its two LDS-DMA requests.  Group B starts one cycle after group A, so the two epilogues never coincide.
// Needs M % 128 == 0 and N % 256 == 0 (row offsets ride in the scalar offset, which the descriptor's range check does not cover), K % 128 == 0,
// K >= 256, N % 16 == 0.  Plain epilogue only (bias, scale); everything else stays on gemm8p.
#include "common.h"
#include "gemm8p.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned h8_u32x4 __attribute__((ext_vector_type(4)));

constexpr int H8_UNIT = 16384;
constexpr int H8_LDS = 8 * H8_UNIT;
constexpr int H8_LDS_ALLOC = H8_LDS + 8 * 1024;       // + one 1 KiB bias line per wave

struct H8Args {
    const bf16* X;
    const bf16* W;
    bf16* Y;
    const bf16* bias;
    int M, N, K;
    int ldx, ldw, ldy;
    float scale;
    int tiles_n, ht_per_col, total;      // tile columns, half-tile rows per column, half-tiles in all (column-major order)
    long long* trace;                    // H8_TRACE builds: [8 waves][128] clock stamps of workgroup 0 (one per cycle: its mode in bit 0-1)
};

#ifndef H8_EPI_PRIO
#define H8_EPI_PRIO 2
#endif
#ifndef H8_TRACE
#define H8_TRACE 0
#endif
#ifndef H8_STORE_AUX
#define H8_STORE_AUX 18
#endif
#define H8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define H8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define H8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int N> __device__ __forceinline__ void h8_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ int h8_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

__global__ __launch_bounds__(512) void gemm8h_kernel(H8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int x = lane & 15, g = lane >> 4;
    const int nk = a.K >> 6;                   // K tiles of 64
    const int C = nk >> 1, P = C + 1;          // cycles (8 phases = 2 K tiles) a group computes per half-tile; period with the epilogue cycle

    // ---- this CU's share of the half-tiles.  XCD j (blocks b % 8 == j) owns an eighth of the half-tile ROWS across all tile columns;
    // inside it the half-tiles are numbered column by column and cut into equal consecutive shares: a CU walks down a column segment,
    // the CUs of an XCD that hold the same rows in different columns read the same X slices at the same time (one fetch into that L2
    // per 8 columns, as gemm8p's 4 x 8 tile groups), a column's W slices are shared by the CUs that hold its segments.
    // (A first version numbered the half-tiles column-major over the whole matrix: an XCD then held one whole column and every XCD
    // read ALL of X -- 5.4 GB for 40960 x 2048 x 8192 -- and the kernel ran at 0.9 of gemm8p.)
    const int G = gridDim.x;
    const bool xs = (G & 7) == 0;
    const int nx = xs ? 8 : 1, Gx = G / nx, xcd = xs ? (int)(blockIdx.x & 7) : 0, cul = xs ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int row_lo = (int)((long long)xcd * a.ht_per_col / nx), row_hi = (int)((long long)(xcd + 1) * a.ht_per_col / nx);
    const int nrows = row_hi - row_lo;                       // half-tile rows of this XCD
    // ... in BANDS of 8 tile columns (a band at a time: the XCD's CUs hold ~4 segments of each of the band's 8 columns, so a column's W
    // slices have four readers in that L2; with all tile columns of a wide output side by side every CU streamed a W column of its own:
    // 40960 x 8192 x 2048 ran at 0.92 of gemm8p, 40960 x 2048 x 2048 at 0.98)
    const int nbands = (a.tiles_n + 7) >> 3;
    int band = 0, first = 0, last = 0;             // the CU's share [first, last) of band `band` (ids: column-major inside the band)
    auto enter_band = [&](int b) {
        const int cols = min(8, a.tiles_n - 8 * b), total = nrows * cols;
        const int per = total / Gx, extra = total % Gx;
        first = cul * per + (cul < extra ? cul : extra);
        last = first + per + (cul < extra ? 1 : 0);
    };
    enter_band(0);

    // ---- staging offsets: unit type 0 Xa, 1 Wb, 2 Xb, 3 Wa; each wave moves pieces `wave` (LDS rows of group A for X units) and
    // `wave + 8` (group B's): rows are relative to the group's half-tile (X) / the tile column (W).  One lane offset per operand and
    // piece: the second unit of an operand (Xb: 64 rows on, Wb: 32 rows on) is a SCALAR offset of the request (this kernel has no
    // register to spare: gemm8p keeps all eight lane offsets).  The scalar offset is outside the descriptor's range check, hence
    // M % 128 == 0 and N % 256 == 0 here: every row a request can name exists.
    int voffx[2], voffw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (wave + 8 * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((lr >> 1) & 7);
        const int tt = (lr >> 4) & 1, xp = lr & 15;
        voffx[i] = (lr & 63) * a.ldx * 2 + c * 16;
        voffw[i] = ((lr >> 5) * 64 + 8 * (xp >> 2) + 4 * tt + (xp & 3)) * a.ldw * 2 + c * 16;
    }
    const int xb_off = 64 * a.ldx * 2, wb_off = 32 * a.ldw * 2;

    // ---- fragment addressing (gemm8p's)
    const int s0 = g ^ ((x >> 1) & 7);
    const int o0 = x * 128 + (s0 << 4), o1 = x * 128 + ((s0 ^ 4) << 4);
    const char* bx0 = smem + wr * 8192 + o0;
    const char* bx1 = smem + wr * 8192 + o1;
    const char* bw0 = smem + wc * 4096 + o0;
    const char* bw1 = smem + wc * 4096 + o1;
    auto rdX = [&](bf16x8 (&f)[2][4], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[0][j] = *(const bf16x8*)(bx0 + slot * H8_UNIT + j * 2048);
            f[1][j] = *(const bf16x8*)(bx1 + slot * H8_UNIT + j * 2048);
        }
    };
    auto rdW = [&](bf16x8 (&f)[2][2], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f[0][t] = *(const bf16x8*)(bw0 + slot * H8_UNIT + t * 2048);
            f[1][t] = *(const bf16x8*)(bw1 + slot * H8_UNIT + t * 2048);
        }
    };

    f32x4 acc[4][8];
    bf16x8 fx[2][4], fwA[2][2], fwB[2][2];

#define H8_MM(FX, FW, J0, T0, I0, I1)                                                          \
    do {                                                                                       \
        _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_) {                               \
            const int ks_ = i_ >> 3, j_ = (i_ >> 1) & 3, t_ = i_ & 1;                          \
            mma16(acc[(T0) + t_][(J0) + j_], FW[ks_][t_], FX[ks_][j_]);                        \
        }                                                                                      \
    } while (0)

    const float scale = a.scale;
    const bool has_scale = scale != 1.f;
#if H8_TRACE
    int tr_n = 0;
#endif

    for (;;) {
        if (first >= last) {
            if (++band >= nbands) break;
            enter_band(band);
            continue;
        }
        // ---- one run: half-tiles first .. first + n - 1 of tile column `col`
        const int col = 8 * band + first / nrows, r0l = first % nrows, r0 = row_lo + r0l;
        const int n = min(last - first, nrows - r0l);
        first += n;
        const int n0 = col * 256;
        // descriptors of the run: X rows from the run's first row to M (the group's half-tile offset rides in the scalar offset), W
        // rows n0 .. N, Y rows from the run's first row
        auto mk_desc = [&](const bf16* base, long long row0, long long rows, int ld) {
            long long rem = (rows - row0) * ld * 2;
            if (rem > 0xffffffffLL) rem = 0xffffffffLL;
            if (rem < 0) rem = 0;
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)row0 * ld), 0, (int)(unsigned)rem, 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t dX = mk_desc(a.X, (long long)r0 * 128, a.M, a.ldx);
        const __amdgpu_buffer_rsrc_t dW = mk_desc(a.W, n0, a.N, a.ldw);
        const __amdgpu_buffer_rsrc_t dY = mk_desc(a.Y, (long long)r0 * 128, a.M, a.ldy);
        const __amdgpu_buffer_rsrc_t dBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 2 : 0, 0x00020000);
        const int xstep = 128 * a.ldx * 2, ystep = 128 * a.ldy * 2;         // bytes from one half-tile of the run to the next

        // group state as a function of the cycle: A starts at cycle 0 with item 0, B at cycle 1 with item 1; per item C compute cycles,
        // then one epilogue cycle.  st_*: phase inside the period (0 .. C - 1 compute, C epilogue), it_*: item, -1 = not started
        // Every wave tracks BOTH groups (it moves rows of both); the schedule is a pure function of the cycle: no communication.
        int phA = 0, itA = 0, phB = P - 1, itB = -1;                        // B idles through cycle 0: the last cycle of a period before its item 1
        auto computing = [&](int ph, int it) { return ph >= 0 && ph < C && it < n; };
        auto advance = [&](int& ph, int& it) {
            ++ph;
            if (ph == P) { ph = 0; it += 2; }
        };
        // cycles of the run: until both groups have stored their last item
        const int nA = (n + 1) >> 1, nB = n >> 1;
        const int cycles = max(nA * P, nB ? 1 + nB * P : 0);

        // stage one unit: K tile kk (already wrapped) of the stream; X units: piece `wave` = group A's rows, `wave + 8` = group B's
        auto stage = [&](int ty, int slot, int kbyte, int rowA, int rowB) __attribute__((always_inline)) {
            if (ty & 1) {
                const int so = kbyte + (ty == 1 ? wb_off : 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(dW, (lds_void*)(smem + slot * H8_UNIT + (wave + 8 * i) * 1024), 16, voffw[i], so, 0, 0);
            } else {
                const int so = kbyte + (ty == 2 ? xb_off : 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dX, (lds_void*)(smem + slot * H8_UNIT + wave * 1024), 16, voffx[0], so + rowA, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dX, (lds_void*)(smem + slot * H8_UNIT + (wave + 8) * 1024), 16, voffx[1], so + rowB, 0, 0);
            }
        };

        // ---- prologue: units -1 .. 5 of the stream  (Wa(0) | Xa(0) Wb(0) Xb(0) Wa(1) | Xa(1) Wb(1)); group A's item 0, group B idle
        // (its rows of the first cycle are never read: it reads row offset 0 too)
        stage(3, 7, 0, 0, 0);
        stage(0, 0, 0, 0, 0);
        stage(1, 1, 0, 0, 0);
        stage(2, 2, 0, 0, 0);
        stage(3, 3, 128, 0, 0);
        stage(0, 4, 128, 0, 0);
        stage(1, 5, 128, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] = vzero<f32x4>();
        {   // the wave's 64 bias values into its LDS line (lanes 0-7 carry 16 bytes each, the others deposit zeros; no bias: zeros)
            const int ln = h8_lane();
            const unsigned off = ln < 8 ? (unsigned)((n0 + wc * 64 + ln * 8) * 2) : 0xffffffffu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dBias, (lds_void*)(smem + H8_LDS + wave * 1024), 16, off, 0, 0, 0);
        }
        H8_VMCNT(11);
        H8_BARRIER();
        rdW(fwA, 7);
        H8_LGKM0();
        if (wr) H8_BARRIER();                // waves 4-7 run half a phase behind waves 0-3

        // One sub-step of a group's epilogue: vector (q, j) of the wave's 16: rows 16 (J0 + j) + x of the half-tile, columns
        // n0 + 64 wc + 32 (T0 >> 1) + 8 g ..+7.  `yrow`: byte offset of the half-tile inside the run (scalar offset of the store).
        // The lane constants and the bias values are set up ONCE per epilogue cycle (EpiCtx: in that cycle the group's 64 fragment
        // registers are free); a sub-step is 8 adds, the scale, 4 conversions and the store.
        struct EpiCtx { unsigned off0, off2; f32x4 b[4]; };
        auto epi_ctx = [&]() __attribute__((always_inline)) {
            EpiCtx e;
            const int ln = h8_lane();
            const int ncol_l = wc * 64 + 8 * (ln >> 4);
            const unsigned base = (unsigned)(((ln & 15) * a.ldy + n0 + ncol_l) * 2);
            e.off0 = n0 + ncol_l < a.N ? base : 0xffffffffu;
            e.off2 = n0 + ncol_l + 32 < a.N ? base + 64u : 0xffffffffu;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 br = *(const bf16x8*)(smem + H8_LDS + wave * 1024 + ((ln >> 4) + 4 * t) * 16);
                e.b[2 * t] = f32x4{(float)br[0], (float)br[1], (float)br[2], (float)br[3]};
                e.b[2 * t + 1] = f32x4{(float)br[4], (float)br[5], (float)br[6], (float)br[7]};
            }
            return e;
        };
        const unsigned rstep = (unsigned)(16 * a.ldy * 2);
        auto epi_sub = [&](auto s_c, int yrow, const EpiCtx& e) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value, q = s >> 2, j = s & 3;
            constexpr int J0 = (q == 0 || q == 1) ? 0 : 4, T0 = (q == 0 || q == 3) ? 0 : 2;
            const unsigned ob = T0 ? e.off2 : e.off0;
            const unsigned off = ob == 0xffffffffu ? ob : ob + (unsigned)(J0 + j) * rstep;
            f32x4 lo = acc[T0][J0 + j] + e.b[T0], hi = acc[T0 + 1][J0 + j] + e.b[T0 + 1];
            if (has_scale) { lo *= scale; hi *= scale; }
            const f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(h8_u32x4, __builtin_convertvector(v, bf16x8)), dY, off, yrow, H8_STORE_AUX);
            acc[T0][J0 + j] = vzero<f32x4>();
            acc[T0 + 1][J0 + j] = vzero<f32x4>();
        };

        // A compute phase (gemm8p's): READ this phase's fragments, request the unit read 6 phases from now, counted wait, barrier, MFMAs
#define H8_PHASE(READ, TY, SLOT, KB, RA, RB, FX, FW, J0, T0, WAITN, HOOK)                        \
    do {                                                                                         \
        READ;                                                                                    \
        stage((TY), (SLOT), (KB), (RA), (RB));                                                   \
        h8_vmcnt<(WAITN)>();                                                                     \
        H8_BARRIER();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_setprio(1);                                                           \
        H8_MM(FX, FW, J0, T0, 0, 8);                                                             \
        HOOK;                                                                                    \
        H8_MM(FX, FW, J0, T0, 8, 15);                                                            \
        H8_LGKM0();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        H8_BARRIER();                                                                            \
        H8_MM(FX, FW, J0, T0, 15, 16);                                                           \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)
        // The same phase for a group that is in its epilogue cycle (or idle): the requests and both barriers, no fragments, no MFMAs;
        // sub-steps S0 and S0 + 1 of the epilogue in the two slots (EP: false = idle, nothing to store)
#define H8_EPHASE(TY, SLOT, KB, RA, RB, S0, LAST, WAITN)                                         \
    do {                                                                                         \
        stage((TY), (SLOT), (KB), (RA), (RB));                                                   \
        if (ep) epi_sub(std::integral_constant<int, (S0)>(), yrow, ectx);    \
        if (ep) h8_vmcnt<(WAITN)>();                                                             \
        else H8_VMCNT(10);                                                                       \
        H8_BARRIER();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (ep) epi_sub(std::integral_constant<int, (S0) + 1>(), yrow, ectx); \
        if (LAST) rdW(fwA, 7);                                                                   \
        H8_LGKM0();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        H8_BARRIER();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)

        // Per cycle: row offsets (bytes, inside the run) of both groups' half-tiles for THIS cycle and the NEXT one (units are requested
        // up to 3 K tiles ahead: DK = 1 is this cycle's second K tile, DK = 2, 3 the next cycle's) and the wrapped K offsets.
        int kt = 0;                                   // the stream's K tile at the start of the cycle (wraps at nk; nk is even)
        int rA0, rB0, rA1, rB1, k1, k2, k3;
        // (The K loop of gemm8p has no scalar bookkeeping at all; here every instruction of it sits at the head of a memory cluster, on
        // the critical path: the first version -- both groups' state advanced and tested twice per cycle, ~60 SALU -- made a compute
        // cycle 4850 clocks against gemm8p's 4290.  A group's rows during a cycle in which it does not compute are never read, so the
        // offsets simply follow its item: the current one, and the next one from the last cycle of its period on.)
        // Scalars of a cycle.  What phase 1 needs -- the K offset of this cycle's second K tile and both groups' rows in THIS cycle --
        // is what the previous cycle called k3 / rA1 / rB1: a rename at the head (cycle_head).  The rest (the next cycle's rows, k2, k3,
        // the groups' state) is computed inside phase 1's MFMA cluster (cycle_mid), where scalar instructions are free; at the head
        // of the cycle every one of them sits in a memory cluster, on the critical path (38 SALU there: a compute cycle of 4450
        // clocks against gemm8p's 4290).
        auto cycle_head = [&]() __attribute__((always_inline)) {
#if H8_TRACE
            if (a.trace && blockIdx.x == 0 && tr_n < 128) {
                const long long t = __builtin_readcyclecounter();
                if (lane == 0) a.trace[wave * 128 + tr_n] = t;
                ++tr_n;
            }
#endif
            rA0 = rA1;
            rB0 = rB1;
            k1 = k3;
        };
        auto cycle_mid = [&]() __attribute__((always_inline)) {
            // state -> the next cycle's; rows of the next cycle; K offsets of the next cycle's two K tiles
            advance(phA, itA);
            advance(phB, itB);
            rA1 = (itA < n ? itA : 0) * xstep;
            rB1 = (itB >= 0 && itB < n ? itB : 0) * xstep;
            kt += 2;
            if (kt >= nk) kt -= nk;
            k2 = kt * 128;
            k3 = k2 + 128;
        };
        // Counted waits.  vmcnt is ONE in-order counter for the LDS-DMA requests and the output stores: "everything but the N youngest
        // has completed" must cover the unit the next phase reads (requested 5 phases ago) and should NOT cover the epilogue's stores,
        // whose completion takes thousands of clocks and would hold BOTH groups at the next barrier (measured: with a flat vmcnt(10)
        // the stores cost 7 % at K = 2048).  Younger than that unit: 2 requests per phase since, and the stores of the epilogue phases
        // among the last five: 11 + 2 min(k, 5) in epilogue phase k, 10 + 2 max(0, 5 - k') in phase k' of the compute cycle after it.
        auto compute_cycle = [&](auto relax_c) __attribute__((always_inline)) {
            constexpr bool R = decltype(relax_c)::value;
            // (preparing a cycle's scalars inside the last MFMA cluster of the cycle before it -- where scalar instructions are free --
            // keeps seven more SGPRs alive across the loop edge and tipped the register allocation over: 144 VGPRs spilled; they stay at
            // the head of the cycle, ~38 SALU instructions)
            cycle_head();
            H8_PHASE(rdX(fx, 0), 2, 6, k1, rA0, rB0, fx, fwA, 0, 0, R ? 20 : 10, cycle_mid());
            H8_PHASE(rdW(fwB, 1), 3, 7, k2, 0, 0, fx, fwB, 0, 2, R ? 18 : 10, (void)0);
            H8_PHASE(rdX(fx, 2), 0, 0, k2, rA1, rB1, fx, fwB, 4, 2, R ? 16 : 10, (void)0);
            H8_PHASE(rdW(fwB, 3), 1, 1, k2, 0, 0, fx, fwA, 4, 0, R ? 14 : 10, (void)0);
            H8_PHASE(rdX(fx, 4), 2, 2, k2, rA1, rB1, fx, fwB, 0, 0, R ? 12 : 10, (void)0);
            H8_PHASE(rdW(fwA, 5), 3, 3, k3, 0, 0, fx, fwA, 0, 2, 10, (void)0);
            H8_PHASE(rdX(fx, 6), 0, 4, k3, rA1, rB1, fx, fwA, 4, 2, 10, (void)0);
            H8_PHASE(rdW(fwA, 7), 1, 5, k3, 0, 0, fx, fwB, 4, 0, 10, (void)0);
        };
        // the cycle of a group that does not compute: its epilogue (ep: 16 sub-steps, two per phase) or nothing (start offset, tail)
        auto other_cycle = [&](bool ep, int yrow) __attribute__((always_inline)) {
            cycle_head();
            EpiCtx ectx = epi_ctx();
            // the epilogue's VALU work runs beside the other group's MFMA clusters (s_setprio 1): above them, or it only gets the issue
            // slots they leave (an epilogue cycle then takes 8500 clocks for BOTH groups instead of ~4300)
            if (ep) __builtin_amdgcn_s_setprio(H8_EPI_PRIO);
            H8_EPHASE(2, 6, k1, rA0, rB0, 0, false, 11);
            cycle_mid();
            H8_EPHASE(3, 7, k2, 0, 0, 2, false, 13);
            H8_EPHASE(0, 0, k2, rA1, rB1, 4, false, 15);
            H8_EPHASE(1, 1, k2, 0, 0, 6, false, 17);
            H8_EPHASE(2, 2, k2, rA1, rB1, 8, false, 19);
            H8_EPHASE(3, 3, k3, 0, 0, 10, false, 21);
            H8_EPHASE(0, 4, k3, rA1, rB1, 12, false, 21);
            H8_EPHASE(1, 5, k3, 0, 0, 14, true, 21);
            __builtin_amdgcn_s_setprio(0);
        };
        // This wave's group: [one idle cycle for group B] then per item C compute cycles + the epilogue cycle, then idle cycles until
        // both groups are done.  (Loop nests, no if / else between whole cycles: a diamond with 192 live registers on both arms sent
        // 120 of them to scratch.)
        int done = 0;
        rA1 = 0;                                       // cycle 0: group A on item 0, group B idle (its rows are never read)
        rB1 = 0;
        k3 = 128;
        if (wr) { other_cycle(false, 0); ++done; }
        if (wr < n) {                                  // the group's first half-tile: no stores behind its first cycle
            for (int i = 0; i < C; ++i) compute_cycle(std::false_type());
            other_cycle(true, wr * ystep);
            done += P;
        }
        for (int item = wr + 2; item < n; item += 2) {
            compute_cycle(std::true_type());           // (its first five waits step over the previous epilogue's stores)
            for (int i = 1; i < C; ++i) compute_cycle(std::false_type());
            other_cycle(true, item * ystep);
            done += P;
        }
        for (; done < cycles; ++done) other_cycle(false, 0);
        H8_VMCNT(0);                         // no LDS-DMA may outlive the run (the next run's prologue reuses the ring)
        if (!wr) H8_BARRIER();               // balance the stagger barrier
        H8_BARRIER();                        // every wave is done with the ring and the bias lines
    }
}

}  // namespace

bool gemm8h_supported(int M, int N, int K, int ldx, int ldw, int ldy) {
    return gemm8p_supported(M, N, K, ldx, ldw, ldy) && M % 128 == 0 && N % 256 == 0 && (long long)M * ldy * 2 < 0xffffffffLL;
}

int launch_gemm8h(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, int M, int N, int K, float scale,
                  hipStream_t st) {
    if (!gemm8h_supported(M, N, K, ldx, ldw, ldy)) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm8h: shape M=%d N=%d K=%d not supported", M, N, K);
    H8Args a;
    a.X = X; a.W = W; a.Y = Y; a.bias = bias;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.scale = scale;
    a.tiles_n = cdiv(N, 256);
    a.ht_per_col = M / 128;
    a.total = a.tiles_n * a.ht_per_col;
    a.trace = nullptr;
#if H8_TRACE
    if (const char* e = getenv("MMGL_H8_TRACE")) a.trace = (long long*)strtoull(e, nullptr, 0);
#endif
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm8h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, H8_LDS_ALLOC);
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    const int n_cu = gemm8p_num_cu();
    // (a grid that is a multiple of 8 whenever every XCD gets at least a pair of half-tiles per CU: the XCD-aware numbering needs it)
    const int pairs = (a.total + 1) / 2;
    int grid = pairs < n_cu ? pairs : n_cu;
    if (grid >= 8) grid &= ~7;
    hipLaunchKernelGGL(gemm8h_kernel, dim3(grid), dim3(512), H8_LDS_ALLOC, st, a);
    MMGL_CHECK_LAUNCH("gemm8h");
    return MMGL_OK;
}
