// 128x128-tile NT GEMM for gfx950 (bf16) for outputs the persistent 256x256 kernel cannot fill the chip with:
//     Y[M,N] = (act((X[M,K] . W[N,K]^T + bias) * scale) masked by zmask) + resid
// replaces: the nn.Linear calls of reference model/modelling_cross_attention.py at the reference's OWN batch
//           (language_modelling/run_generation.py:124-126: per_device_train_batch_size 2-4, so M = B * 640 = 2560 rows): out_proj
//           (:273), the gated layers' q_proj / out_proj (:194-199), fc1 (:352), and their dgrads -- 2560 x 2048 is 80 tiles of
//           256x256 on 256 CUs (0.31 rounds) and 2560 x 8192 is 1.25 rounds; in 128x128 tiles they are 1.25 and 5 rounds of two
//           workgroups per CU.
//
// Structure: 256 threads = 4 waves in 2 (m) x 2 (n), each a 64x64 block as 2x2 v_mfma_f32_32x32x16_bf16 accumulators (64 VGPRs).
// TWO workgroups share a CU (64 KiB of LDS each): one's barrier / load wait is the other's MFMA time.
//   * operands stream global -> LDS by LDS-DMA (16 B per lane) in K steps of GM_BK = 64: an X tile and a W tile of 128 rows x 128 B
//     per stage, GM_NS = 2 stages in a ring; stage t + NS - 1 is issued right after the barrier that opens step t, and waited for
//     with a counted vmcnt.  Rows past M / N read as zero through the buffer descriptor (no clamping, no branches).
//   * LDS rows are 128 B with the 16-byte chunk XOR-swizzled by (row >> 1) & 7 on the source side (the DMA destination is
//     lane-linear): every ds_read_b128 of a 32-row fragment is bank-conflict free.
//   * W is the MFMA A operand: a lane ends up with ONE output row (m = lane & 31) and 16 columns per accumulator; one
//     v_permlane32_swap per register pair turns them into runs of 8 consecutive columns: 16-byte loads of bias / zmask / resid,
//     16-byte stores, no LDS round trip.
// Measured (MI355X, tools/probes/gemm_mid.py; us at 2560x2048x2048 / 2560x8192x2048 / 6500x768x768): this configuration 32.7 / 93 /
// 15.5 (657 / 920 / 494 TF; the round-1 128x128 kernel it replaces: 38.4 / - / 20.3; hipBLASLt 30.7 at the first).  A deeper ring does
// NOT help: 32-wide K stages with 5 / 4 / 3 stages in flight (2, 2, 3 workgroups per CU) 39 / 114 / 17.8, one workgroup per CU with
// four 64-wide stages 45 / 117 / 21 -- the kernel is bound by the LDS itself, not by load latency: per K step a workgroup's waves
// read 64 KiB of fragments and the DMA writes 32 KiB for 512 matrix-pipe cycles (the 256x256 kernel: 192 + 64 KiB for 2048), so
// what counts is how much of that the second workgroup overlaps; input row pitches (K + 0 / 64 / 128 / 192 elements) change nothing.
// Needs K % 32 == 0, N % 8 == 0, row strides % 8 == 0.
#include "common.h"
#include "attn_common.h"
#include "gemm8p.h"
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef GM_BK
#define GM_BK 64                     // K elements per stage (64: 128-byte LDS rows, 32: 64-byte rows)
#endif
#ifndef GM_NS
#define GM_NS 2                      // ring stages; GM_NS stages of 256 rows x 2 GM_BK bytes per workgroup
#endif
#ifndef GM_OCC
#define GM_OCC 2                     // workgroups per CU the LDS ring is sized for
#endif
constexpr int GM_ROWB = GM_BK * 2, GM_TILEB = 128 * GM_ROWB, GM_STAGE = 2 * GM_TILEB;
constexpr int GM_DPS = GM_TILEB / 4096 * 2;     // DMA instructions per wave and stage (X + W)
static_assert(GM_NS * GM_STAGE * GM_OCC <= 160 * 1024, "LDS");

struct MidArgs {
    const bf16* X;
    const bf16* W;
    bf16* Y;
    const bf16* bias;
    const bf16* resid;
    const bf16* zmask;
    int M, N, K, ldx, ldw, ldy;
    float scale;
    int act, tiles_m, tiles_n;
};

#define GM_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define GM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void swap32_f32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int ACT> __device__ __forceinline__ float mid_act(float v) {
    if constexpr (ACT == 1) return fmaxf(v, 0.f);
    else if constexpr (ACT == 2) {                                       // erf by Abramowitz-Stegun 7.1.26, as the persistent kernel's epilogue (gemm8p.hip)
        const float z = fabsf(v) * 0.70710678118654752f;
        const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
        const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
        return 0.5f * v * (1.f + copysignf(1.f - poly * __expf(-z * z), v));
    }
    else if constexpr (ACT == 3) return v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
    else if constexpr (ACT == 4) return 0.5f * v * (1.f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
    else return v;
}

template <int N> __device__ __forceinline__ void gm_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(256, GM_OCC) void gemm_mid_kernel(MidArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NS = GM_NS, PD = NS - 1, DPS = GM_DPS;
    constexpr int RPI = 1024 / GM_ROWB;                                  // LDS rows per DMA instruction (8 or 16)
    constexpr int CPR = GM_ROWB / 16;                                    // 16-byte chunks per row (8 or 4)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;

    int tm, tn;
    grouped_tile(xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n), a.tiles_m, a.tiles_n, tm, tn);
    const int m0 = tm * 128, n0 = tn * 128;
    const uint32_t ldxB = (uint32_t)a.ldx * 2u, ldwB = (uint32_t)a.ldw * 2u;
    const int mrows = min(128, a.M - m0), nrows = min(128, a.N - n0);
    // descriptors over this tile's rows: bytes up to the end of the last row's K range.  A contraction padded past x's row length
    // (K > ldx against zero columns of W: mmgl_gemm_nt's contract) reads the head of the next row there, and nothing past the last row.
    uint32_t xbytes = (uint32_t)(mrows - 1) * ldxB + (uint32_t)a.K * 2u;
    if (mrows == a.M - m0) xbytes = min(xbytes, (uint32_t)mrows * ldxB);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.X + (size_t)m0 * a.ldx, xbytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.W + (size_t)n0 * a.ldw, (uint32_t)(nrows - 1) * ldwB + (uint32_t)a.K * 2u);

    // The 16-byte chunk c of LDS row `row` holds source chunk c ^ swz(row): swz = (row >> 1) & 7 for 128-byte rows, (row >> 2) & 3 for
    // 64-byte rows -- the 16 lanes the LDS serves together (rows {0-3, 12-15, 20-27} + 4k of one fragment) then cover all 64 banks.
    // DMA instruction i of this wave fills rows 4 RPI i + RPI wave + lane / CPR of a tile (1 KiB, lane-linear); swz does not depend on i.
    const int drow = RPI * wave + lane / CPR;
    const int dswz = GM_BK == 64 ? (drow >> 1) & 7 : (drow >> 2) & 3;
    const uint32_t dchunk = (uint32_t)(((lane & (CPR - 1)) ^ dswz) * 16);
    const uint32_t xvoff = (uint32_t)drow * ldxB + dchunk, wvoff = (uint32_t)drow * ldwB + dchunk;
    const int nk = a.K / GM_BK;
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        char* base = smem + slot * GM_STAGE + wave * 1024;
        const int k0 = t * GM_ROWB;                                      // bytes
#pragma unroll
        for (int i = 0; i < DPS / 2; ++i) {
            // the row-block advance rides in the VECTOR offset: only voffset (+ the immediate) is range-checked against the descriptor,
            // a scalar offset is not -- with it there, rows past M / N of a ragged tile were READ (up to 127 rows beyond the operand)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(base + i * 4096), 16, xvoff + (uint32_t)(i * 4 * RPI) * ldxB, k0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + GM_TILEB + i * 4096), 16, wvoff + (uint32_t)(i * 4 * RPI) * ldwB, k0, 0, 0);
        }
    };
#pragma unroll
    for (int t = 0; t < PD; ++t) issue(t < nk ? t : 0, t);              // (unconditional: exact vmcnt counts)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read: row l31 of a 32-row block, K slice ks (16 wide): chunk 2 ks + hi, swizzled
    const int swz = GM_BK == 64 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;
    const int xrow = (wr * 64 + l31) * GM_ROWB, wrow = GM_TILEB + (wc * 64 + l31) * GM_ROWB;
    int slot = 0, islot = PD % NS;
    for (int t = 0; t < nk; ++t) {
        const int ahead = nk - 1 - t;                                    // stages after t that were issued: min(ahead, PD - 1)
        if (PD >= 4 && ahead >= 3) gm_vmcnt<3 * DPS>();
        else if (PD >= 3 && ahead >= 2) gm_vmcnt<2 * DPS>();
        else if (PD >= 2 && ahead >= 1) gm_vmcnt<DPS>();
        else gm_vmcnt<0>();
        GM_BARRIER();
        if (t + PD < nk) issue(t + PD, islot);
        const char* s = smem + slot * GM_STAGE;
#pragma unroll
        for (int ks = 0; ks < GM_BK / 16; ++ks) {
            const int ch = ((2 * ks + hi) ^ swz) * 16;
            const bf16x8 w0 = *(const bf16x8*)(s + wrow + ch), w1 = *(const bf16x8*)(s + wrow + 32 * GM_ROWB + ch);
            const bf16x8 x0 = *(const bf16x8*)(s + xrow + ch), x1 = *(const bf16x8*)(s + xrow + 32 * GM_ROWB + ch);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[1][1], 0, 0, 0);
        }
        slot = (slot + 1 == NS) ? 0 : slot + 1;
        islot = (islot + 1 == NS) ? 0 : islot + 1;
    }

    // ---- epilogue.  acc[nb][mb][r]: row m = m0 + 64 wr + 32 mb + l31, column n0 + 64 wc + 32 nb + (r & 3) + 8 (r >> 2) + 4 hi.
    // After the half-wave swap of registers 4c..4c+3 with 4c+4..4c+7 (c = 0, 2) a lane holds columns 8 (c + hi) .. + 7 in order.
    const uint32_t ldyB = (uint32_t)a.ldy * 2u;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(a.bias, a.bias ? (uint32_t)a.N * 2u : 0u);
    const uint32_t ybytes = (uint32_t)(mrows - 1) * ldyB + (uint32_t)a.N * 2u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.Y + (size_t)m0 * a.ldy, ybytes);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.resid ? a.resid + (size_t)m0 * a.ldy : nullptr, a.resid ? ybytes : 0u);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.zmask ? a.zmask + (size_t)m0 * a.ldy : nullptr, a.zmask ? ybytes : 0u);
    const bool has_z = a.zmask != nullptr, has_r = a.resid != nullptr;
    const float scale = a.scale;
    auto finish = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const int n = n0 + 64 * wc + 32 * nb + 8 * (c + hi);
                const bool n_ok = n < a.N;
                const f32x8 bv = __builtin_convertvector(buf_load8<bf16>(rb, n_ok ? (uint32_t)n * 2u : OOB), f32x8);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const int ml = 64 * wr + 32 * mb + l31;
                    const uint32_t off = (n_ok && ml < mrows) ? (uint32_t)ml * ldyB + (uint32_t)n * 2u : OOB;
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = acc[nb][mb][4 * c + r]; v[4 + r] = acc[nb][mb][4 * c + 4 + r]; }
#pragma unroll
                    for (int r = 0; r < 4; ++r) swap32_f32(v[r], v[4 + r]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = mid_act<ACT>((v[e] + bv[e]) * scale);
                    if (has_z) {
                        const f32x8 z = __builtin_convertvector(buf_load8<bf16>(rz, off), f32x8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = z[e] > 0.f ? v[e] : 0.f;
                    }
                    if (has_r) {
                        const f32x8 rs = __builtin_convertvector(buf_load8<bf16>(rr, off), f32x8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rs[e];
                    }
                    const f32x8 o8 = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, __builtin_convertvector(o8, bf16x8)), ry, off, 0, 0);
                }
            }
    };
    switch (a.act) {
        case 1: finish(std::integral_constant<int, 1>()); break;
        case 2: finish(std::integral_constant<int, 2>()); break;
        case 3: finish(std::integral_constant<int, 3>()); break;
        case 4: finish(std::integral_constant<int, 4>()); break;
        default: finish(std::integral_constant<int, 0>()); break;
    }
}

}  // namespace

bool gemm_mid_supported(int M, int N, int K, int ldx, int ldw, int ldy) {
    return M > 0 && N > 0 && K >= GM_BK && K % GM_BK == 0 && N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 && ldx + 127 >= K && ldw >= K && ldy >= N &&
           (long long)ldx * 2 * 128 < 0x7fffffffLL && (long long)ldw * 2 * 128 < 0x7fffffffLL && (long long)ldy * 2 * 128 < 0x7fffffffLL;
}

int launch_gemm_mid(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, const bf16* resid, const bf16* zmask,
                    int M, int N, int K, int act, float scale, hipStream_t st) {
    if (!gemm_mid_supported(M, N, K, ldx, ldw, ldy))
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm_mid: shape M=%d N=%d K=%d (ld %d %d %d) not supported", M, N, K, ldx, ldw, ldy);
    MidArgs a;
    a.X = X; a.W = W; a.Y = Y; a.bias = bias; a.resid = resid; a.zmask = zmask;
    a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.scale = scale; a.act = act;
    a.tiles_m = cdiv(M, 128); a.tiles_n = cdiv(N, 128);
    constexpr int LDS = GM_NS * GM_STAGE;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_mid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute(gemm_mid): %s", hipGetErrorString(e));
        configured = true;
    }
    hipLaunchKernelGGL(gemm_mid_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), LDS, st, a);
    MMGL_CHECK_LAUNCH("gemm_mid");
    return MMGL_OK;
}
