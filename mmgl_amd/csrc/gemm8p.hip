// Persistent 256x256 "ping-pong" NT GEMM for gfx950 (bf16):   Y[M,N] = epi( X[M,K] . W[N,K]^T )
// replaces: every nn.Linear of the frozen path of reference model/modelling_cross_attention.py -- q/k/v/out_proj of the
//           frozen decoder layers (:194-199, :273), fc1+ReLU / fc2 (:352-355), lm_head (:826), the RoBERTa / CLIP encoder
//           linears behind :992 / :1018 -- and their dgrads (dx = dy . W as an NT GEMM against W^T).
//
// Structure (one workgroup = 8 waves = one CU, persistent over output tiles):
//   * wave (wr, wc) = (wave >> 2, wave & 3) owns a 128 (m) x 64 (n) block of the 256x256 tile: 128 fp32 accumulator VGPRs.
//     W is the MFMA A operand with its fragment rows mapped to n = 32 (t >> 1) + 8 g' + 4 (t & 1) + r  (a free row
//     permutation in the staging addresses), so a lane ends up with 8 CONSECUTIVE output columns per fragment-row pair and the
//     four lanes of an output row with 64 contiguous bytes: one 16-byte store per lane, whole 64-byte segments per instruction
//     (the first mapping, 16 g' + 4 t + r, wrote 16-byte pieces at a 32-byte pitch: ~9 us of store drain per tile),
//     bias / activation / residual without cross-lane traffic.
//   * the two waves that share a SIMD (w and w + 4) run half a phase apart: a phase is  [ds_read fragments | issue LDS-DMA |
//     s_waitcnt vmcnt(N)] s_barrier [16 MFMA] s_barrier ; waves 4-7 take one extra barrier up front, so while one wave of
//     a SIMD is in its MFMA cluster the other one is in its memory cluster (and the matrix pipe never waits for LDS).
//   * operands stream global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane) in UNITS of 128 LDS rows x 128 B
//     (64 bf16 of K) = 16 KiB, the exact set of rows ONE phase reads: Xa (rows 0-63 of each wave's m block), Wb (fragment
//     rows t = 2,3), Xb (rows 64-127), Wa (t = 0,1 of the NEXT K tile: read one phase early into the W register set the
//     finished Wb just freed, so every phase reads 8 / 4 / 8 / 4 fragments and no phase is LDS-bound).  8 unit slots of LDS
//     (128 KiB) form a ring: the unit read in phase p sits in slot p & 7, was issued in phase p - 6 and waited for (counted
//     vmcnt, never 0) in phase p - 1; its slot is re-issued in phase p + 2.  The stream of units runs across K tiles AND
//     across output tiles: the first
//     K tiles of the next output tile are in flight while the current tile finishes and stores, so there is no per-tile
//     pipeline fill.  Rows are addressed through buffer descriptors rebuilt per tile (rows past M / N read as zero: no
//     clamping, no branches), the K offset rides in the scalar offset: no vector address arithmetic in the loop.
//   * LDS rows are 128 B with the 16-byte slot XOR-swizzled by (row >> 1) & 7 on the SOURCE side (an LDS-DMA destination is
//     lane-linear): every ds_read_b128 of a fragment (16 consecutive rows, same k slot) is bank-conflict free.
// Needs K % 128 == 0, K >= 256, N % 16 == 0.  Everything else goes to the 128x128 kernel of gemm.hip.
#include "common.h"
#include "gemm8p.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned p8_u32x4 __attribute__((ext_vector_type(4)));

constexpr int P8_UNIT = 16384;      // bytes per staged unit
constexpr int P8_LDS = 8 * P8_UNIT;
constexpr int P8_LDS_TOTAL = P8_LDS + 16 * 1024;     // + two 1 KiB bias lines per wave (tile parity)
constexpr int P8_LDS_ALLOC = P8_LDS_TOTAL + 64;      // + the dynamic schedule's mailbox

struct P8Args {
    const bf16* X;
    const bf16* W;
    bf16* Y;
    const bf16* bias;      // [N] or null
    const bf16* resid;     // [M,N] or null: added after the activation
    const bf16* zmask;     // [M,N] or null: output zeroed where zmask <= 0 (ReLU backward of the tensor this GEMM differentiates)
    int M, N, K;
    int ldx, ldw, ldy;     // row strides in elements (multiples of 8); resid and zmask share ldy
    float scale;
    int act;
    int tiles_m, tiles_n, total;
    int nsplit;            // K splits per output tile (1 = none): work item = (tile, split), fp32 partial tiles into `part`
    float* part;           // [nsplit][M][N] fp32 (ACT = 5 kernels)
    unsigned* bits_out;    // ACT = 1 kernels: one bit per output element (y > 0), [tile][wave][lane][4 dwords], or NULL
    const unsigned* bits_in;   // ZR kernels: the same bits as the mask of this output (instead of a zmask tensor), or NULL
    int tile0;             // first output tile of this launch (a hybrid launch: full tiles first, the rest as K-split items)
    long long* trace;      // timing experiments only (P8_TRACE builds): [8 waves][64] clock stamps of workgroup trace_wg
    int trace_wg;
    unsigned* sched;       // dynamic tile schedule (mmgl_gemm_set_tile_counter): [0..7] per-XCD item counters, [8] finished workgroups;
                           // all zero between launches.  NULL: static schedule (work item it * gridDim + wg)
};

// Work items of a launch in the order XCD j takes them (j = blockIdx & 7 when the grid is a multiple of 8, else one list): item k
// of list j is virtual id (k / Gx) * G + j * Gx + k % Gx -- round by round the XCD's 8 x 4 group of tiles (grouped_tile), exactly
// the static schedule's assignment.  The DYNAMIC schedule hands ALL items out through one atomic counter per list: a workgroup
// whose CU is shared with (or was held back by) another kernel -- RCCL's all-reduce during the backward pass -- simply takes fewer
// of them, or none, instead of finishing its statically assigned tiles late while 255 CUs wait (the round-2 measurement: any
// co-resident kernel stretched a GEMM by 38 %; with only the later rounds dynamic the displaced workgroups' first tiles still ran
// last: +11 %).  A list that runs dry is refilled from the next XCD's (a steal costs the L2 locality of that tile only).
// Returns the virtual id, or -1 when every list is empty.
__device__ __forceinline__ int p8_fetch_item(unsigned* sched, int xcd, int nlists, int G, int Gx, int nitems) {
    for (int t = 0; t < nlists; ++t) {
        const int j = xcd + t < nlists ? xcd + t : xcd + t - nlists;
        const unsigned k = atomicAdd(&sched[j], 1u);
        const long long v = (long long)(k / (unsigned)Gx) * G + j * Gx + (int)(k % (unsigned)Gx);
        if (v < nitems) return (int)v;
    }
    return -1;
}

// ACT is a compile-time family: 0 = none, 1 = ReLU, 2 gelu (erf), 3 quick_gelu, 4 gelu (tanh)
template <int ACT> __device__ __forceinline__ float p8_act(float v) {
    if constexpr (ACT == 2) {
        // exact-GELU with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 ulp): ~14 VALU per value instead
        // of libm's erff -- the epilogue runs beside the partner wave's MFMAs and must stay short
        const float z = fabsf(v) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);     // v_rcp_f32 (1 ulp); __frcp_rn / operator/ expand to the 10-instruction IEEE division
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        const float er = 1.f - poly * __expf(-z * z);
        return 0.5f * v * (1.f + copysignf(er, v));
    }
    else if constexpr (ACT == 3) return v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
    else if constexpr (ACT == 4) {
        const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
        return 0.5f * v * (1.f + tanhf(u));
    } else if constexpr (ACT == 1) return fmaxf(v, 0.f);
    else return v;
}

// lane id from the exec mask (v_mbcnt): recomputed where it is needed instead of keeping a register (or, under this kernel's
// register pressure, a scratch slot and its vmcnt(0) reload) alive across the main loop
// (asm volatile: the builtin form is loop-invariant to hipcc, which hoists it -- and everything derived from it -- to kernel entry)
__device__ __forceinline__ int p8_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// Cache policy bits of the output stores (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).  nt + sc1: the output tile is streamed out without
// staying in the L2 that holds the operand tiles the other workgroups of the XCD are about to read -- 40960x8192x2048 1029 -> 920 us
// (1336 -> 1494 TF), 6144 columns 754 -> 706, 2048 columns 269 -> 253, the seven shapes of tools/probes/gemm_step_shapes.py 3494 -> 3299 us;
// in the training step (whose next kernel reads that output) 256.3 -> 259.8 samples/s.  0 / 1 / 16 / 17: no change; 2 / 3: as 18 but
// 2 % behind on the K = 768 shape.
#ifndef P8_STORE_AUX
#define P8_STORE_AUX 18
#endif
#ifndef P8_ZMASK_AUX
#define P8_ZMASK_AUX 2               // ... of the zmask tile loads of the epilogue
#endif
#ifndef P8_PART_AUX
#define P8_PART_AUX 0                // ... of the K-split scratch tiles (read back by the finish kernel right away)
#endif
#ifndef P8_LOADX_AUX
#define P8_LOADX_AUX 0               // cache policy bits of the operand streams (timing experiments)
#endif
#ifndef P8_LOADW_AUX
#define P8_LOADW_AUX 0
#endif
#ifndef P8_TRACE
#define P8_TRACE 0                   // timing experiments only: clock stamps of workgroup 0 into P8Args::trace
#endif
#define P8_BARRIER() asm volatile("s_barrier" ::: "memory")
#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int ACT, bool ZR> __global__ __launch_bounds__(512) void gemm8p_kernel(P8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int x = lane & 15, g = lane >> 4;
    int nk = a.K >> 6;                         // 64-wide K tiles of the current work item (varies by item under a K split)

    // ---- persistent tile schedule: in every round of gridDim.x tiles, XCD j (blocks b % 8 == j) takes 32 consecutive
    // virtual ids = one 8 x 4 group of tiles (grouped_tile): activation K slices are shared by 4 workgroups of one L2
    const int G = gridDim.x;
    const int wg = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    // work item -> (tile origin, first K element, K tiles): with a K split the K range is cut into nsplit runs of whole
    // 128-wide steps (the first K/128 % nsplit runs one step longer)
    const bool dyn = a.sched != nullptr;
    const int nlists = (G & 7) == 0 ? 8 : 1, Gx = G / nlists, xcd = nlists == 8 ? (int)(blockIdx.x & 7) : 0;
    int* mailbox = (int*)(smem + P8_LDS_TOTAL);           // dynamic schedule: the item after next, by tile parity (written a whole tile ahead)
    auto item_origin = [&](int v, int& m0, int& n0, int& k0, int& nki, int& sp) -> bool {
        if (v < 0 || v >= a.total * a.nsplit) return false;
        int tm, tn;
        sp = v % a.nsplit;
        grouped_tile(a.tile0 + v / a.nsplit, a.tiles_m, a.tiles_n, tm, tn);
        m0 = tm * 256;
        n0 = tn * 256;
        const int units = a.K >> 7, base = units / a.nsplit, rem = units % a.nsplit;
        k0 = (sp * base + (sp < rem ? sp : rem)) * 128;
        nki = (base + (sp < rem ? 1 : 0)) * 2;
        return true;
    };
    auto tile_origin = [&](int it, int& m0, int& n0, int& k0, int& nki, int& sp) -> bool { return item_origin(it * G + wg, m0, n0, k0, nki, sp); };
    // rows [row0, rows) of an operand with row stride ld: everything past the last row reads as zero.  (A K tile may run past
    // the end of a ROW when the caller pads K -- lm_head's dgrad contracts over V = 50272 -- and then reads the start of the next
    // row: finite values that meet the zero padding of the other operand.)
    auto mk_desc = [&](const bf16* base, int row0, int rows, int ld, int k0, bool valid) {
        long long rem = valid ? (long long)(rows - row0) * ld * 2 - (long long)k0 * 2 : 0;
        if (rem > 0xffffffffLL) rem = 0xffffffffLL;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (valid ? (size_t)row0 * ld + k0 : 0)), 0, (int)(unsigned)rem, 0x00020000);
    };

    // ---- staging offsets: unit type 0 Xa, 1 Wb, 2 Xb, 3 Wa; each wave moves pieces `wave` and `wave + 8` (8 LDS rows each)
    int voff[4][2];
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (wave + 8 * i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((lr >> 1) & 7);
            int srow;
            if (!(ty & 1)) srow = (lr >> 6) * 128 + (ty == 2 ? 64 : 0) + (lr & 63);
            else {
                const int u = ty == 1 ? 1 : 0, tt = (lr >> 4) & 1, xp = lr & 15;
                srow = (lr >> 5) * 64 + 32 * u + 8 * (xp >> 2) + 4 * tt + (xp & 3);      // n = 32 (t >> 1) + 8 g + 4 (t & 1) + r
            }
            voff[ty][i] = srow * ((ty & 1) ? a.ldw : a.ldx) * 2 + c * 16;
        }
    auto stage = [&](int ty, int slot, __amdgpu_buffer_rsrc_t rs, int kbyte) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ty & 1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * P8_UNIT + (wave + 8 * i) * 1024), 16, voff[ty][i],
                                                         kbyte, 0, P8_LOADW_AUX);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * P8_UNIT + (wave + 8 * i) * 1024), 16, voff[ty][i],
                                                         kbyte, 0, P8_LOADX_AUX);
        }
    };

    // ---- fragment addressing: row x of a 16-row block, k slot (ks * 4 + g) ^ key
    const int s0 = g ^ ((x >> 1) & 7);
    const int o0 = x * 128 + (s0 << 4), o1 = x * 128 + ((s0 ^ 4) << 4);
    const char* bx0 = smem + wr * 8192 + o0;
    const char* bx1 = smem + wr * 8192 + o1;
    const char* bw0 = smem + wc * 4096 + o0;
    const char* bw1 = smem + wc * 4096 + o1;
    auto rdX = [&](bf16x8 (&f)[2][4], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[0][j] = *(const bf16x8*)(bx0 + slot * P8_UNIT + j * 2048);
            f[1][j] = *(const bf16x8*)(bx1 + slot * P8_UNIT + j * 2048);
        }
    };
    auto rdW = [&](bf16x8 (&f)[2][2], int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f[0][t] = *(const bf16x8*)(bw0 + slot * P8_UNIT + t * 2048);
            f[1][t] = *(const bf16x8*)(bw1 + slot * P8_UNIT + t * 2048);
        }
    };

    f32x4 acc[4][8];
#if P8_TRACE
    int tr_n = 0;
    auto stamp = [&](int it_) __attribute__((always_inline)) {
        if (a.trace && (int)blockIdx.x == a.trace_wg && it_ >= 2 && tr_n < 64) {
            const long long t = __builtin_readcyclecounter();
            if (lane == 0) a.trace[wave * 64 + tr_n] = t;      // a plain store: perturbs vmcnt a little, same for every variant compared
            ++tr_n;
        }
    };
#define P8_STAMP(it_) stamp(it_)
#else
#define P8_STAMP(it_) (void)0
#endif

    // the 16 MFMAs of a phase, MFMAs [I0, I1) of the order (ks, j, t)
#define P8_MM(FX, FW, J0, T0, I0, I1)                                                          \
    do {                                                                                       \
        _Pragma("unroll") for (int i_ = (I0); i_ < (I1); ++i_) {                               \
            const int ks_ = i_ >> 3, j_ = (i_ >> 1) & 3, t_ = i_ & 1;                          \
            mma16(acc[(T0) + t_][(J0) + j_], FW[ks_][t_], FX[ks_][j_]);                        \
        }                                                                                      \
    } while (0)

    int m0 = 0, n0 = 0, m1 = 0, n1 = 0, k0 = 0, k1 = 0, nk1 = 0, sp0 = 0, sp1 = 0;
    int it = 0;
    int vcur = wg, vnext = G + wg;                        // virtual ids of the current / next work item (static schedule: it * G + wg)
    if (dyn) {                                            // the first item: one lane asks, everybody waits (~1 us, once per launch)
        if (tid == 0) mailbox[2] = p8_fetch_item(a.sched, xcd, nlists, G, Gx, a.total * a.nsplit);
        __syncthreads();
        vcur = mailbox[2];
        if (vcur < 0) {                                   // nothing left for this workgroup (it started late: its CU was busy)
            if (tid == 0 && atomicAdd(&a.sched[8], 1u) == (unsigned)G - 1u)
                for (int i = 0; i < 9; ++i) atomicExch(&a.sched[i], 0u);
            return;
        }
    }
    if (!item_origin(vcur, m0, n0, k0, nk, sp0)) return;  // (static: grid <= items, every workgroup has its round-0 item)
    bool have_next = dyn ? false : tile_origin(1, m1, n1, k1, nk1, sp1);    // dynamic: known after the prologue's barrier
    __amdgpu_buffer_rsrc_t dXc = mk_desc(a.X, m0, a.M, a.ldx, k0, true), dWc = mk_desc(a.W, n0, a.N, a.ldw, k0, true);
    __amdgpu_buffer_rsrc_t dXn = mk_desc(a.X, m1, a.M, a.ldx, k1, have_next), dWn = mk_desc(a.W, n1, a.N, a.ldw, k1, have_next);

    // ---- epilogue plumbing.  Lane (x, g) owns, for each of its 8 rows m = m0 + wr*128 + 16 J + x, the columns
    // n = n0 + wc*64 + 32 (t >> 1) + 8 g + 4 (t & 1) + r.  The lane's 16 bias values are fetched at the top of the tile's last K-tile pair, five
    // phases before their first use, so that the wait for them is a counted vmcnt behind younger LDS-DMA loads, not a drain.
    // Output rows go through a buffer descriptor based at row m0 (rows past M and, via an all-ones offset, columns past N are
    // dropped by the hardware: no branches around the stores).
    // The wave's 64 bias values travel by LDS-DMA into a wave-private 1 KiB line past the unit ring (lanes 0-7 carry 16 bytes
    // each, the others point outside the descriptor and deposit zeros): no register holds them across the tile.  The descriptor
    // covers bias[0..N) (empty without a bias): columns past N and the no-bias case read as zero, branch-free.
    const __amdgpu_buffer_rsrc_t dBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 2 : 0, 0x00020000);
    // two lines per wave, by tile parity: with the spread epilogue a tile's line is read during the NEXT tile's first phases
    auto fetch_bias = [&](int nt0, int par) __attribute__((always_inline)) {
        const int ln = p8_lane();
        const unsigned off = ln < 8 ? (unsigned)((nt0 + wc * 64 + ln * 8) * 2) : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dBias, (lds_void*)(smem + P8_LDS + (par & 1) * 8192 + wave * 1024), 16, off, 0, 0, 0);
    };
    auto y_desc = [&](const bf16* base, int mt0) {
        long long rem = (long long)(a.M - mt0) * a.ldy * 2;
        if (rem > 0xffffffffLL) rem = 0xffffffffLL;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)mt0 * a.ldy), 0, (int)(unsigned)rem, 0x00020000);
    };
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != 1.f;       // wave-uniform: whole loops are skipped
    // Epilogue of a tile.  Lane geometry is recomputed here (it must not live across the main loop), the bias / scale passes
    // are whole-quadrant loops of packed fp32 ops behind uniform branches, and plain ReLU is one packed integer max on the
    // converted bf16 pairs (a negative bf16 is a negative int16).  ZR kernels request the 16 zmask (else residual) vectors of
    // the tile up front: one exposed memory latency per tile instead of one per quadrant.
    auto epilogue_q = [&](int m0, int n0, int par, int q_lo, int q_hi) __attribute__((always_inline)) {
        if constexpr (ACT == 5) {
            // K-split work item: the raw fp32 accumulators of this split go to its own 256 x 256 fp32 scratch tile,
            // part[(tile - tile0) * nsplit + split][256][256] (whole tiles: no bounds to check); bias / activation / masks are applied
            // by p8_splitk_finish_kernel when it folds the splits
            const int ln = p8_lane();
            const int ncol_l = wc * 64 + 8 * (ln >> 4);
            const unsigned base = (unsigned)(((wr * 128 + (ln & 15)) * 256 + ncol_l) * 4);
            const unsigned rstep = (unsigned)(16 * 256 * 4);
            const int vitem = vcur;                                          // = (tile - tile0) * nsplit + split
            const __amdgpu_buffer_rsrc_t dP = __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + ((size_t)vitem << 16)), 0, 256 * 256 * 4, 0x00020000);
#pragma unroll
            for (int T0 = 0; T0 < 4; T0 += 2)
#pragma unroll
                for (int J = 0; J < 8; ++J) {
                    const unsigned off = base + (unsigned)J * rstep + (unsigned)(64 * T0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p8_u32x4, acc[T0][J]), dP, off, 0, P8_PART_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(p8_u32x4, acc[T0 + 1][J]), dP, off + 16, 0, P8_PART_AUX);
                    acc[T0][J] = vzero<f32x4>();
                    acc[T0 + 1][J] = vzero<f32x4>();
                }
            return;
        }
        const int ln = p8_lane();
        const int ncol_l = wc * 64 + 8 * (ln >> 4);          // first column of this lane inside the tile (fragment rows 0, 1)
        const bool col_ok0 = n0 + ncol_l < a.N, col_ok1 = n0 + ncol_l + 32 < a.N;
        const unsigned base = (unsigned)(((wr * 128 + (ln & 15)) * a.ldy + n0 + ncol_l) * 2);
        const unsigned rstep = (unsigned)(16 * a.ldy * 2);
        auto y_off = [&](int J, int T0) __attribute__((always_inline)) -> unsigned {
            return (T0 ? col_ok1 : col_ok0) ? base + (unsigned)J * rstep + (unsigned)(32 * T0) : 0xffffffffu;
        };
        const __amdgpu_buffer_rsrc_t dY = y_desc(a.Y, m0);
        bf16x8 pre[16];
        p8_u32x4 mbits = {0u, 0u, 0u, 0u};
        if constexpr (ZR) {
            const __amdgpu_buffer_rsrc_t dZ = y_desc(a.zmask ? a.zmask : a.resid, m0);
            if (a.bits_in) {                         // 16 bytes per lane instead of 16 x 16: the ReLU mask as bits (see bits_out)
                mbits = *(const p8_u32x4*)(a.bits_in + ((size_t)((a.tile0 + vcur) * 8 + wave) * 64 + ln) * 4);
            } else if (a.zmask) {                           // read once, never again: non-temporal (-2.5 % on the fc2 dgrad); a residual tile is
#pragma unroll                                       // the stream the next kernels read too: default policy (nt: +1.5 %)
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int J0 = (q == 0 || q == 1) ? 0 : 4, T0 = (q == 0 || q == 3) ? 0 : 2;
                        pre[q * 4 + j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(dZ, y_off(J0 + j, T0), 0, P8_ZMASK_AUX));
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int J0 = (q == 0 || q == 1) ? 0 : 4, T0 = (q == 0 || q == 3) ? 0 : 2;
                        pre[q * 4 + j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(dZ, y_off(J0 + j, T0), 0, 0));
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < q_lo || q >= q_hi) continue;
            const int J0 = (q == 0 || q == 1) ? 0 : 4, T0 = (q == 0 || q == 3) ? 0 : 2;
            if (has_bias) {
                const bf16x8 br = *(const bf16x8*)(smem + P8_LDS + (par & 1) * 8192 + wave * 1024 + ((ln >> 4) + 2 * T0) * 16);     // columns 32 (T0 / 2) + 8 g ..
                const f32x4 b0 = {(float)br[0], (float)br[1], (float)br[2], (float)br[3]}, b1 = {(float)br[4], (float)br[5], (float)br[6], (float)br[7]};
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[T0][J0 + j] += b0; acc[T0 + 1][J0 + j] += b1; }
            }
            if (has_scale) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[T0][J0 + j] *= a.scale; acc[T0 + 1][J0 + j] *= a.scale; }
            }
            unsigned qbits = 0;                      // ACT = 1: (y > 0) bits of this quadrant, byte j = vector j
            // MODE (wave-uniform, ZR kernels): 1 = zmask only (applied to the packed bf16 output), 2 = residual only, 3 = both,
            // 4 = mask bits only
            auto rows = [&](auto mode_tag) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned off = y_off(J0 + j, T0);
                    const f32x4 lo = acc[T0][J0 + j], hi = acc[T0 + 1][J0 + j];
                    f32x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    if constexpr (ACT >= 2 || (ACT == 1 && ZR)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = p8_act<ACT>(v[e]);
                    }
                    if constexpr (MODE == 2) {
                        const bf16x8 z = pre[q * 4 + j];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)z[e];
                    }
                    if constexpr (MODE == 3) {
                        const bf16x8 z = pre[q * 4 + j];
                        const __amdgpu_buffer_rsrc_t dR = y_desc(a.resid, m0);
                        const bf16x8 rr = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(dR, off, 0, 0));
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (((float)z[e] > 0.f) ? v[e] : 0.f) + (float)rr[e];
                    }
                    p8_u32x4 ob = __builtin_bit_cast(p8_u32x4, __builtin_convertvector(v, bf16x8));
                    if constexpr (ACT == 1 && !ZR) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned w = ob[e];
                            asm("v_pk_max_i16 %0, %1, 0" : "=v"(w) : "v"(w));     // ReLU on two bf16: a negative bf16 is a negative int16
                            ob[e] = w;
                        }
                        if (a.bits_out) {                                        // wave-uniform
                            unsigned by = 0;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                unsigned t = ob[e];
                                asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(t) : "v"(t));      // each half: 0 / 1 (after the ReLU: > 0 <=> != 0)
                                t |= t >> 15;                                                           // bit 0 = low half, bit 1 = high half (bit 16: garbage)
                                by |= t << (2 * e);
                            }
                            qbits |= (by & 0xffu) << (8 * j);
                        }
                    }
                    if constexpr (MODE == 1) {
                        // zmask > 0 on the raw bf16 pairs: negatives -> 0, positives -> 1, then 0 - that = 0xffff / 0 per half
                        const p8_u32x4 zz = __builtin_bit_cast(p8_u32x4, pre[q * 4 + j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned m = zz[e];
                            asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]\n\tv_pk_sub_u16 %0, 0, %0" : "=v"(m) : "v"(m));
                            ob[e] &= m;
                        }
                    }
                    if constexpr (MODE == 4) {
                        const unsigned by = mbits[q] >> (8 * j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            unsigned t = (by >> (2 * e)) & 3u;
                            t = (t | (t << 15)) & 0x00010001u;                   // the two bits, one per half
                            unsigned mk;
                            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(mk) : "v"(t), "v"(0xffffu));      // 0 / 0xffff per half (full-rate multiply)
                            ob[e] &= mk;
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(ob, dY, off, 0, P8_STORE_AUX);
                }
            };
            if constexpr (!ZR) rows(std::integral_constant<int, 0>());
            else if (a.bits_in) rows(std::integral_constant<int, 4>());
            else if (a.zmask && a.resid) rows(std::integral_constant<int, 3>());
            else if (a.zmask) rows(std::integral_constant<int, 1>());
            else rows(std::integral_constant<int, 2>());
            if constexpr (ACT == 1 && !ZR) {
                if (a.bits_out) a.bits_out[((size_t)((a.tile0 + vcur) * 8 + wave) * 64 + ln) * 4 + q] = qbits;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[T0][J0 + j] = vzero<f32x4>(); acc[T0 + 1][J0 + j] = vzero<f32x4>(); }
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) { epilogue_q(m0, n0, it, 0, 4); };

    // (Start offsets between groups of CUs -- a quarter of a tile apart, so that the groups' output bursts do not collide -- paid for
    // themselves in round 2 (1047 -> 1021 us at 40960x8192x2048) and stopped doing so with the streamed output stores: re-measured in
    // round 3 for outputs of >= 10 rounds of tiles, groups inside an XCD 946 -> 950-1029 us, groups of whole XCDs 934 -> 935-948 us
    // (a gross gain of ~3 % against a tail of 2.5-4.4 %).  Removed.)

    // ---- prologue: units -1 .. 5 of the stream  (Wa(0) | Xa(0) Wb(0) Xb(0) Wa(1) | Xa(1) Wb(1))
    stage(3, 7, dWc, 0);
    stage(0, 0, dXc, 0);
    stage(1, 1, dWc, 0);
    stage(2, 2, dXc, 0);
    stage(3, 3, dWc, 128);
    stage(0, 4, dXc, 128);
    stage(1, 5, dWc, 128);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = vzero<f32x4>();
    // two W fragment sets that swap roles every K tile: even K tiles keep Wa in fwA and Wb in fwB, odd ones Wa in fwB, Wb in fwA
    bf16x8 fx[2][4], fwA[2][2], fwB[2][2];
    fetch_bias(n0, 0);                   // the lane's 16 bias values of the first tile
    if (dyn && tid == 0) {               // the next two work items of this workgroup (one lane asks, the mailbox tells the other waves)
        // A workgroup never holds more items up front than the static schedule would give it, ceil(items / G): with three claims per
        // workgroup before any work is done, an output of one round of tiles was computed by a third of the CUs (3 tile times instead
        // of 1) and one of two rounds by 128 workgroups with three tiles + 128 with one (3 instead of 2).  From three rounds up
        // the claims are as before: two in the prologue, then one per finished tile.
        const int nit = a.total * a.nsplit, cap = (nit + G - 1) / G;
        const int i1 = cap >= 2 ? p8_fetch_item(a.sched, xcd, nlists, G, Gx, nit) : -1;
        mailbox[0] = i1;
        mailbox[1] = (cap >= 3 && i1 >= 0) ? p8_fetch_item(a.sched, xcd, nlists, G, Gx, nit) : -1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    P8_VMCNT(11);                        // units -1 and 0 have landed
    P8_BARRIER();
    rdW(fwA, 7);
    P8_LGKM0();
    if (wr) P8_BARRIER();                // waves 4-7 run half a phase behind waves 0-3
    if (dyn) {
        vnext = mailbox[0];
        have_next = item_origin(vnext, m1, n1, k1, nk1, sp1);
        dXn = mk_desc(a.X, m1, a.M, a.ldx, k1, have_next);
        dWn = mk_desc(a.W, n1, a.N, a.ldw, k1, have_next);
    }

    // One phase.  READ: this phase's fragments.  (TY, SLOT, DK): the unit staged now = the one read 6 phases from now, of
    // K tile kt + DK (past the end of this output tile: K tile kt + DK - nk of the next one).  vmcnt(10): the 5 younger units
    // may stay in flight.  MMA: the 16 MFMAs.  The memory cluster is the critical path of the ping-pong (two LDS-DMA issues and
    // up to 8 ds_reads against the partner's 16 MFMAs): nothing else lives in this loop -- no branch, no address arithmetic.
#define P8_PHASE_M(READ, TY, SLOT, DK, FX, FW, J0, T0, MM, WAITN)                                \
    do {                                                                                         \
        READ;                                                                                    \
        {                                                                                        \
            const int kk_ = kt + (DK);                                                           \
            const bool nx_ = kk_ >= nk;                                                          \
            const int kb_ = (nx_ ? kk_ - nk : kk_) * 128;                                        \
            if ((TY) & 1) stage((TY), (SLOT), nx_ ? dWn : dWc, kb_);                             \
            else stage((TY), (SLOT), nx_ ? dXn : dXc, kb_);                                      \
        }                                                                                        \
        P8_VMCNT(WAITN);                                                                         \
        P8_BARRIER();                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_setprio(1);                                                           \
        /* the barrier that hands the matrix pipe to the partner wave sits BEFORE this wave's last MFMA: the partner's first      */ \
        /* MFMAs queue behind it instead of behind a drained pipe (+1-2 %; two or more MFMAs after the barrier lose 7 %)          */ \
        MM(FX, FW, J0, T0, 0, 15);                                                               \
        P8_LGKM0();                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        P8_BARRIER();                                                                            \
        MM(FX, FW, J0, T0, 15, 16);                                                              \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)

#define P8_PHASE(READ, TY, SLOT, DK, FX, FW, J0, T0) P8_PHASE_M(READ, TY, SLOT, DK, FX, FW, J0, T0, P8_MM, 10)
    // (Relaxed counts -- vmcnt(27) -- in a tile's first five phases, so that they do not wait for the previous tile's 16 output stores:
    // measured neutral in rounds 2 and 3 -- the phases run at full speed, the epilogue stretches by as much -- and removed.)
    for (;;) {
        P8_STAMP(it);
        for (int kt = 0; kt < nk; kt += 2) {
            P8_PHASE(rdX(fx, 0), 2, 6, 1, fx, fwA, 0, 0);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdW(fwB, 1), 3, 7, 2, fx, fwB, 0, 2);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdX(fx, 2), 0, 0, 2, fx, fwB, 4, 2);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdW(fwB, 3), 1, 1, 2, fx, fwA, 4, 0);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdX(fx, 4), 2, 2, 2, fx, fwB, 0, 0);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdW(fwA, 5), 3, 3, 3, fx, fwA, 0, 2);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdX(fx, 6), 0, 4, 3, fx, fwA, 4, 2);
            if (kt < 4) P8_STAMP(it);
            P8_PHASE(rdW(fwA, 7), 1, 5, 3, fx, fwB, 4, 0);
            if (kt < 4) P8_STAMP(it);
        }
        P8_STAMP(it);
        {
            // The two wave groups reach this point half a phase apart, and whichever runs its epilogue holds the other at its
            // next barrier: left alone, the two epilogues (and tile switches) run one after the other (~13k clocks per tile in
            // the clock stamps).  One extra barrier each re-aligns them: waves 0-3 take theirs before the epilogue (it pairs with
            // the barrier inside waves 4-7's last MFMA cluster), waves 4-7 after the tile switch (it pairs with the first barrier
            // of waves 0-3's next phase) -- both groups then do their epilogues between the same two barriers, side by side.
            if (!wr) P8_BARRIER();
            // dynamic schedule: one lane asks for the item after the next one now -- the answer comes back under the epilogue -- and
            // leaves it in the mailbox slot of this tile's parity, which every wave reads at the NEXT tile switch (a whole tile and
            // dozens of barriers from now; the other slot is the one being read at this switch)
            // (only when the tile after the next one exists -- slot (it + 1) & 1, written a switch ago by this same lane: a fetch whose
            // slot is never read again would take an item out of the lists for good)
            int fetched = -1;
            if (dyn && tid == 0 && have_next && mailbox[(it + 1) & 1] >= 0) fetched = p8_fetch_item(a.sched, xcd, nlists, G, Gx, a.total * a.nsplit);
            epilogue();
            if (dyn && tid == 0) mailbox[it & 1] = fetched;
            P8_STAMP(it);
            if (!have_next) { if (wr) P8_BARRIER(); break; }
        }
        ++it;
        m0 = m1;
        n0 = n1;
        nk = nk1;
        sp0 = sp1;
        dXc = dXn;
        dWc = dWn;
        vcur = vnext;
        vnext = dyn ? mailbox[it & 1] : (it + 1) * G + wg;       // the slot written at the PREVIOUS switch (or by the prologue)
        have_next = item_origin(vnext, m1, n1, k1, nk1, sp1);
        dXn = mk_desc(a.X, m1, a.M, a.ldx, k1, have_next);
        dWn = mk_desc(a.W, n1, a.N, a.ldw, k1, have_next);
        fetch_bias(n0, it);              // the next tile's bias values: consumed by its epilogue a whole tile from now
        if (wr) P8_BARRIER();
    }
    P8_VMCNT(0);                         // no LDS-DMA may outlive the workgroup
    if (!wr) P8_BARRIER();               // balance the stagger barrier
    if (dyn && tid == 0) {
        // every workgroup has stopped asking by now; the last one to leave puts the counters back to zero for the next launch
        if (atomicAdd(&a.sched[8], 1u) == (unsigned)G - 1u) {
#pragma unroll
            for (int i = 0; i < 9; ++i) atomicExch(&a.sched[i], 0u);
        }
    }
}

// fold the K splits of ACT = 5 work items and apply the epilogue:  Y = act((sum_s part[s] + bias) * scale) (zmask, + resid)
__global__ __launch_bounds__(256) void p8_splitk_finish_kernel(const float* __restrict__ part, bf16* __restrict__ Y, const bf16* __restrict__ bias,
                                                               const bf16* __restrict__ resid, const bf16* __restrict__ zmask, int M, int N, int ldy,
                                                               int nsplit, int act, float scale, int tile0, int ntiles, int tiles_m, int tiles_n) {
    // one 8-column vector per thread and trip: tile (i >> 13), row (i >> 5) & 255, columns 8 (i & 31) of the split tiles' scratch
    const size_t nv = (size_t)ntiles << 13;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const int tl = (int)(i >> 13), r = (int)(i >> 5) & 255, c8 = ((int)i & 31) * 8;
        int tm, tn;
        grouped_tile(tile0 + tl, tiles_m, tiles_n, tm, tn);
        const size_t m = (size_t)tm * 256 + r;
        const int n = tn * 256 + c8;
        if (m >= (size_t)M || n >= N) continue;
        const float* p = part + (((size_t)tl * nsplit) << 16) + r * 256 + c8;
        f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
        for (int sp = 1; sp < nsplit; ++sp) {
            lo += *(const f32x4*)(p + ((size_t)sp << 16));
            hi += *(const f32x4*)(p + ((size_t)sp << 16) + 4);
        }
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (bias) {
            const bf16x8 b = *(const bf16x8*)(bias + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)b[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = v[e] * scale;
            switch (act) {
                case 1: t = p8_act<1>(t); break;
                case 2: t = p8_act<2>(t); break;
                case 3: t = p8_act<3>(t); break;
                case 4: t = p8_act<4>(t); break;
                default: break;
            }
            v[e] = t;
        }
        const size_t yo = m * (size_t)ldy + n;
        if (zmask) {
            const bf16x8 z = *(const bf16x8*)(zmask + yo);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ((float)z[e] > 0.f) ? v[e] : 0.f;
        }
        if (resid) {
            const bf16x8 r = *(const bf16x8*)(resid + yo);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
        }
        f32x8 o8 = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
        *(bf16x8*)(Y + yo) = __builtin_convertvector(o8, bf16x8);
    }
}

}  // namespace

// Work plan of one GEMM: tiles [0, direct) run as whole 256x256 output tiles, tiles [direct, total) as nsplit K-split work items
// each (fp32 scratch tiles, folded by p8_splitk_finish_kernel).  Two cases split:
//  * an output with too few tiles to fill the chip (direct = 0): about one work item per CU, each at least two 128-wide K steps;
//    measured: a gain from K = 3072 up (K = 8192: 129 -> 92 us at M = 2560), a loss at K <= 2048;
//  * an output of one to three rounds of tiles plus a remainder of at most half a round (the reference's batch: 2560 x 8192 =
//    320 tiles on 256 CUs): the remainder would occupy a quarter of the chip for a whole tile time; cut into K splits it takes a
//    quarter of that.
// N % 8 == 0 is implied by gemm8p_supported.
static int p8_num_cu() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return n;
}
int gemm8p_num_cu() { return p8_num_cu(); }
void gemm8p_plan(int M, int N, int K, int* direct, int* nsplit) {
    const int tiles = cdiv(M, 256) * cdiv(N, 256), units = K / 128;
    *direct = tiles;
    *nsplit = 0;
    if (tiles < 160) {
        constexpr int min_units = 24;                     // K >= 3072 (tools: 2560 x 2048 x K sweep)
        if (units < min_units) return;
        // K in [3072, 4096) with >= 64 tiles (config 2's fc2 / fc1-dgrad at B = 16: 10240 x 768 x 3072 = 120 tiles): two K splits of
        // 12 units each plus the fp32 fold (66.6 + 13.1 us) lose to the 128x128 kernel, whose 480 tiles fill 94 % of its slots:
        // config 2 at B = 16 946 -> 963 samples/s, B = 4 / B = 64 unchanged (round 6, tools/probes/ab_bench.sh)
        if (units < 32 && tiles >= 64) return;
        int s = (224 + tiles - 1) / tiles;
        if (s > 8) s = 8;
        if (s > units / 2) s = units / 2;
        if (s >= 2) { *direct = 0; *nsplit = s; }
        return;
    }
    constexpr int hybrid = 3;                             // smallest split count worth it
    // up to 12 whole rounds (round 4: Llama's 17408-row outputs with N = 4096 are 1088 tiles = 4.25 rounds -- o_proj, down_proj and,
    // in the backward pass, the dgrads of q|k|v and gate|up, K up to 22016: a fifth round at a quarter of the chip cost 15 % of each)
    const int G = p8_num_cu(), rounds = tiles / G, r = tiles % G;
    if (!hybrid || rounds < 1 || rounds > 12 || r == 0) return;
    int s = G / r;
    if (s > 8) s = 8;
    if (s > units / 4) s = units / 4;                     // at least four 128-wide K steps per item
    if (s >= hybrid && s >= 2) { *direct = tiles - r; *nsplit = s; }
}
int gemm8p_splits(int M, int N, int K) {
    int d, s;
    gemm8p_plan(M, N, K, &d, &s);
    return s;
}

bool gemm8p_supported(int M, int N, int K, int ldx, int ldw, int ldy) {
    return M > 0 && N > 0 && K >= 256 && K % 128 == 0 && N % 16 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 &&
           (long long)M * ldx * 2 < 0xffffffffLL && (long long)N * ldw * 2 < 0xffffffffLL;
}

size_t gemm8p_split_bytes(int M, int N, int K) {
    int d, sp;
    gemm8p_plan(M, N, K, &d, &sp);
    return sp ? (size_t)(cdiv(M, 256) * cdiv(N, 256) - d) * sp * (256 * 256 * sizeof(float)) : 0;
}

// Tile counters per DEVICE (mmgl_gemm_set_tile_counter); NULL = static tile schedule.  Process-wide, not per thread: the GEMMs that
// overlap a collective are the backward ones, and those are launched from autograd's device thread, not from the thread that
// switched the schedule on (a thread_local pointer left exactly those launches on the static schedule).
static std::atomic<unsigned*> g_tile_counter[64];
static std::atomic<int> g_tile_counters_set{0};
// The one counter block of a device serves ONE stream: two launches in flight on different streams would hand out each other's
// tiles.  The first launch after mmgl_gemm_set_tile_counter binds the device's counters to its stream; a launch on any other stream
// while they are set runs on the STATIC schedule (it never touches the counters: eval / prefetch / encoder work on a side stream
// stays legal, it just does not yield tiles to a co-resident collective).  MMGL_GEMM_STRICT_STREAM=1 turns that case into an error.
static hipStream_t const P8_STREAM_UNBOUND = (hipStream_t)(intptr_t)-1;
static std::atomic<hipStream_t> g_tile_stream[64];
static unsigned* p8_tile_counter() {
    if (!g_tile_counters_set.load(std::memory_order_relaxed)) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return g_tile_counter[dev & 63].load(std::memory_order_acquire);
}
// counters for a launch on stream `st` (NULL: static schedule); false: the device's counters belong to another stream
static bool p8_tile_counter_for(hipStream_t st, unsigned** out) {
    *out = nullptr;
    if (!g_tile_counters_set.load(std::memory_order_relaxed)) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    unsigned* c = g_tile_counter[dev & 63].load(std::memory_order_acquire);
    if (!c) return true;
    hipStream_t bound = P8_STREAM_UNBOUND;
    if (!g_tile_stream[dev & 63].compare_exchange_strong(bound, st, std::memory_order_acq_rel) && bound != st) return false;
    *out = c;
    return true;
}

int launch_gemm8p(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, const bf16* resid,
                  const bf16* zmask, int M, int N, int K, int act, float scale, hipStream_t st, float* part, size_t part_bytes,
                  unsigned* bits_out, const unsigned* bits_in) {
    if (!gemm8p_supported(M, N, K, ldx, ldw, ldy))
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm8p: shape M=%d N=%d K=%d (ld %d %d %d) not supported", M, N, K, ldx, ldw, ldy);
    P8Args a;
    a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.resid = resid; a.zmask = zmask;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.scale = scale; a.act = act;
    a.tiles_m = cdiv(M, 256); a.tiles_n = cdiv(N, 256); a.total = a.tiles_m * a.tiles_n;
    a.nsplit = 1;
    a.part = nullptr;
    a.bits_out = bits_out;
    a.bits_in = bits_in;
    if (bits_out && (act != 1 || resid || zmask)) MMGL_FAIL(MMGL_ERR_INVALID, "gemm8p: mask bits are written by the plain ReLU epilogue only");
    if (bits_in && (zmask || resid || act != 0)) MMGL_FAIL(MMGL_ERR_INVALID, "gemm8p: mask bits are applied by the plain epilogue only");
    if (bits_out || bits_in) part = nullptr;             // whole tiles only: the bits are indexed by (tile, wave, lane)
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess)
            MMGL_FAIL(MMGL_ERR_HIP, "gemm8p: hipGetDeviceProperties failed");
        n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        const void* ks[9] = {(const void*)gemm8p_kernel<0, false>, (const void*)gemm8p_kernel<0, true>, (const void*)gemm8p_kernel<1, false>,
                             (const void*)gemm8p_kernel<1, true>,  (const void*)gemm8p_kernel<2, false>, (const void*)gemm8p_kernel<3, false>,
                             (const void*)gemm8p_kernel<3, true>,  (const void*)gemm8p_kernel<4, true>,  (const void*)gemm8p_kernel<5, false>};
        for (const void* kf : ks) {
            hipError_t e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_ALLOC);
            if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
    }
    // the work plan (gemm8p_plan): whole tiles [0, direct), then K-split work items for the rest with fp32 scratch tiles in the
    // CALLER's memory (gemm8p_split_bytes), folded -- with the whole epilogue -- by p8_splitk_finish_kernel.  Without scratch
    // nothing is split.
    const int tiles = a.total;
    int direct = tiles, nsplit = 0;
    if (part && part_bytes >= gemm8p_split_bytes(M, N, K)) gemm8p_plan(M, N, K, &direct, &nsplit);
    a.trace = nullptr;
    a.trace_wg = 0;
    a.tile0 = 0;
    if (!p8_tile_counter_for(st, &a.sched)) {
        static const bool strict = getenv("MMGL_GEMM_STRICT_STREAM") && atoi(getenv("MMGL_GEMM_STRICT_STREAM"));
        if (strict)
            MMGL_FAIL(MMGL_ERR_INVALID, "gemm8p: the dynamic tile schedule of this device is bound to another stream (one stream per device "
                                        "while mmgl_gemm_set_tile_counter is in effect)");
        a.sched = nullptr;                               // static schedule for this launch
    }
    if (direct > 0) {
        a.total = direct;
        const int grid = direct < n_cu ? direct : n_cu;
#if P8_TRACE
        if (const char* e = getenv("MMGL_P8_TRACE")) a.trace = (long long*)strtoull(e, nullptr, 0);
        if (const char* e = getenv("MMGL_P8_TRACE_WG")) a.trace_wg = atoi(e);
#endif
        const bool zr = resid || zmask || bits_in;
#define P8_LAUNCH(A, Z) hipLaunchKernelGGL((gemm8p_kernel<A, Z>), dim3(grid), dim3(512), P8_LDS_ALLOC, st, a)
        switch (act) {
            case 0: if (zr) P8_LAUNCH(0, true); else P8_LAUNCH(0, false); break;
            case 1: if (zr) P8_LAUNCH(1, true); else P8_LAUNCH(1, false); break;
            case 2: if (zr) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm8p: gelu(erf) with residual / zmask is not instantiated"); P8_LAUNCH(2, false); break;
            case 3: if (zr) P8_LAUNCH(3, true); else P8_LAUNCH(3, false); break;
            case 4: P8_LAUNCH(4, true); break;
            default: MMGL_FAIL(MMGL_ERR_INVALID, "gemm8p: unknown activation %d", act);
        }
#undef P8_LAUNCH
        MMGL_CHECK_LAUNCH("gemm8p");
    }
    if (nsplit) {
        MMGL_CHECK_ARG(act >= 0 && act <= 4, "gemm8p: unknown activation %d", act);
        const int rest = tiles - direct;
        a.tile0 = direct;
        a.total = rest;
        a.nsplit = nsplit;
        a.part = part;
        a.trace = nullptr;
        const int items = rest * nsplit, g = items < n_cu ? items : n_cu;
        hipLaunchKernelGGL((gemm8p_kernel<5, false>), dim3(g), dim3(512), P8_LDS_ALLOC, st, a);
        MMGL_CHECK_LAUNCH("gemm8p (K split)");
        int blocks = rest * 32;                                  // 8192 vectors per tile, 256 per block and trip
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(p8_splitk_finish_kernel, dim3(blocks), dim3(256), 0, st, part, Y, bias, resid, zmask, M, N, ldy, nsplit, act, scale,
                           direct, rest, a.tiles_m, a.tiles_n);
        MMGL_CHECK_LAUNCH("gemm8p split-K finish");
    }
    return MMGL_OK;
}

extern "C" int mmgl_gemm_set_tile_counter(void* counter) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "mmgl_gemm_set_tile_counter: hipGetDevice: %s", hipGetErrorString(e));
    g_tile_stream[dev & 63].store(P8_STREAM_UNBOUND, std::memory_order_release);       // the next launch binds the counters to its stream
    unsigned* old = g_tile_counter[dev & 63].exchange((unsigned*)counter, std::memory_order_acq_rel);
    g_tile_counters_set.fetch_add((counter != nullptr) - (old != nullptr), std::memory_order_relaxed);
    return MMGL_OK;
}

extern "C" void* mmgl_gemm_get_tile_counter(void) { return p8_tile_counter(); }
