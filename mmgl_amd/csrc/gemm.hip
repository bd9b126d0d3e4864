// Dense projections of the gated cross-attention block on MFMA (gfx950):
//   y = act((x W^T + bias) * scale)  [+ second operand pair for the fused LoRA term],  dgrad, wgrad.
// replaces nn.Linear q/k/v/out_proj, fc1(+ReLU), fc2 of reference model/modelling_cross_attention.py:194-199,273,352-355
// and peft's LoRA linear (model/modelling_self_attention.py:80-87).
//
// One kernel: "NT" GEMM  C[M,N] = X[M,K] . W[N,K]^T  -- both operands contraction-contiguous, which is exactly
// torch's nn.Linear layout, so forward needs no transposes.  128x128 block tile, K step = 128 bytes per row
// (64 bf16 / 32 f32), 4 waves as 2x2, each wave 64x64 = 4x4 v_mfma 16x16 tiles (bf16: 16x16x32; f32: 8 x 16x16x4,
// an exact fp32 fma chain).  Operands are staged global -> VGPR -> LDS with the next tile's loads issued before
// the current tile's MFMAs (double-buffered LDS, one barrier per K tile); LDS rows are 128 B with the 16-B slot
// XOR-swizzled by (row>>1)&7 so a fragment read (16 lanes = 16 rows, same k-slot) is bank-conflict free.
// W is the MFMA A operand and X the B operand, so each lane ends up with 4 CONSECUTIVE output columns of one
// output row: bias add, activation and the 8/16-byte stores need no cross-lane traffic.
// dgrad / wgrad are the same kernel on explicitly transposed operands (tile transpose kernel below); the ReLU
// mask, the out_scale and the bias column-sum are fused into that transpose / mask pass.
#include "common.h"
#include "gemm8p.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128;         // tile rows (x side), tile rows (W side), bytes per LDS row
constexpr int TILE_BYTES = BM * ROWB;                 // 16 KiB per operand per buffer

template <typename T> struct GT {
    static constexpr int VN = 16 / sizeof(T);         // elements per 16-B chunk
    static constexpr int BK = ROWB / sizeof(T);       // 64 bf16 / 32 f32
    static constexpr int KSTEPS = BK / 32;            // 32-wide MFMA contraction steps per tile
    typedef T chunk_t __attribute__((ext_vector_type(16 / sizeof(T))));
};

__device__ __forceinline__ int swz(int row, int c) { return row * ROWB + ((c ^ ((row >> 1) & 7)) << 4); }

template <typename T> __device__ __forceinline__ typename Elem<T>::v8 lds_frag(const char* tile, int row, int ks, int g);
template <> __device__ __forceinline__ bf16x8 lds_frag<bf16>(const char* tile, int row, int ks, int g) {
    return *(const bf16x8*)(tile + swz(row, ks * 4 + g));
}
template <> __device__ __forceinline__ f32x8 lds_frag<float>(const char* tile, int row, int ks, int g) {
    const f32x4 a = *(const f32x4*)(tile + swz(row, 2 * g));
    const f32x4 b = *(const f32x4*)(tile + swz(row, 2 * g + 1));
    f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return r;
}

// GLDS = operand tiles go global -> LDS directly (global_load_lds_dwordx4, 1 KiB = 8 swizzled rows per wave instruction,
// no VGPR round trip, no ds_write pass); the XOR swizzle is applied to the SOURCE chunk index because the LDS
// destination of an LDS-DMA is lane-linear.  Needs K % BK == 0 (no zero fill possible); row tails are clamped to a valid row
// (their outputs are never stored).  The register-staged form remains for ragged K.
template <typename T, int ACT, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ X, const T* __restrict__ W, T* __restrict__ Y,
                                                      const T* __restrict__ bias, int M, int N, int K, float scale,
                                                      int accumulate, const T* __restrict__ X2, const T* __restrict__ W2,
                                                      int K2, int tiles_m, int tiles_n) {
    typedef GT<T> G;
    typedef typename G::chunk_t chunk_t;
    typedef typename Elem<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sX = smem;                       // [2][TILE_BYTES]
    char* sW = smem + 2 * TILE_BYTES;      // [2][TILE_BYTES]

    // XCD-aware tile order: consecutive virtual ids share the W panel (same tile_n) on one XCD's L2
    const int vid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tn = vid / tiles_m, tm = vid % tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = vzero<f32x4>();

    const int nk1 = (K + G::BK - 1) / G::BK;
    const int nk2 = X2 ? (K2 + G::BK - 1) / G::BK : 0;
    const int nk = nk1 + nk2;

    chunk_t rx[4], rw[4];
    auto load_tile = [&](int kt) {
        const T* xs = X; const T* ws = W; int kk = K, k0 = kt * G::BK;
        if (kt >= nk1) { xs = X2; ws = W2; kk = K2; k0 = (kt - nk1) * G::BK; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256, row = id >> 3, c = id & 7;
            const int kc = k0 + c * G::VN;
            rx[i] = (m0 + row < M && kc < kk) ? *(const chunk_t*)(xs + (size_t)(m0 + row) * kk + kc) : vzero<chunk_t>();
            rw[i] = (n0 + row < N && kc < kk) ? *(const chunk_t*)(ws + (size_t)(n0 + row) * kk + kc) : vzero<chunk_t>();
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256, row = id >> 3, c = id & 7;
            *(chunk_t*)(sX + buf * TILE_BYTES + swz(row, c)) = rx[i];
            *(chunk_t*)(sW + buf * TILE_BYTES + swz(row, c)) = rw[i];
        }
    };

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds_tile = [&](int kt, int buf) {
        const T* xs = X; const T* ws = W; int kk = K, k0 = kt * G::BK;
        if (kt >= nk1) { xs = X2; ws = W2; kk = K2; k0 = (kt - nk1) * G::BK; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = (i * 4 + wave_u) * 8;
            const int row = rbase + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const T* gx = xs + (size_t)min(m0 + row, M - 1) * kk + k0 + c * G::VN;
            const T* gw = ws + (size_t)min(n0 + row, N - 1) * kk + k0 + c * G::VN;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gx,
                                             (__attribute__((address_space(3))) void*)(sX + buf * TILE_BYTES + rbase * ROWB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gw,
                                             (__attribute__((address_space(3))) void*)(sW + buf * TILE_BYTES + rbase * ROWB), 16, 0, 0);
        }
    };

    if constexpr (GLDS) glds_tile(0, 0);
    else { load_tile(0); store_tile(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            if constexpr (GLDS) glds_tile(kt + 1, buf ^ 1);
            else load_tile(kt + 1);
        }
        const char* tX = sX + buf * TILE_BYTES;
        const char* tW = sW + buf * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            v8 fw[4], fx[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fw[i] = lds_frag<T>(tW, wn * 64 + i * 16 + x, ks, g);
                fx[i] = lds_frag<T>(tX, wm * 64 + i * 16 + x, ks, g);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mma16(acc[i][j], fw[i], fx[j]);
        }
        if constexpr (!GLDS) { if (kt + 1 < nk) store_tile(buf ^ 1); }
        __syncthreads();
    }

    // epilogue: lane (x = output row within tile j, g) holds columns n = 16 i + 4 g + 0..3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + x;
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + g * 4;
            if (n >= N) continue;
            f32x4 v = acc[i][j];
            if (bias) {
                const typename Elem<T>::v4 bv = *(const typename Elem<T>::v4*)(bias + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
            }
            v *= scale;
            if (ACT == MMGL_ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            T* yp = Y + (size_t)m * N + n;
            if (accumulate) {
                const typename Elem<T>::v4 ov = *(const typename Elem<T>::v4*)yp;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)ov[r];
            }
            if constexpr (sizeof(T) == 2) *(bf16x4*)yp = __builtin_convertvector(v, bf16x4);
            else *(f32x4*)yp = v;
        }
    }
}

// shape test of the backward planners: a dgrad whose output fills the chip twice with 256x256 tiles runs as an NT GEMM against a
// transposed copy of the weight (the persistent kernel of gemm8p.hip), smaller ones on the transposition-free 128x128 kernel
inline bool big_tile_shape(int M, int N, int K) { return K % 64 == 0 && (size_t)cdiv(M, 256) * cdiv(N, 256) >= 512; }

// the persistent ping-pong kernel (gemm8p.hip) takes every bf16 shape it supports; the 256x256 / 128x128 kernels of this file are
// what the remaining shapes run on (K not a multiple of 128, accumulating outputs, fp32, few tiles)
inline constexpr int tune_gemm_8p() { return 1; }
// few-tile shapes (>= 24 tiles: below that even 8 splits leave most of the chip idle and the 128x128 kernel's 4x finer tiles win)
inline bool gemm8p_use_splits(int M, int N, int K) {
    return cdiv(M, 256) * cdiv(N, 256) >= 24 && gemm8p_splits(M, N, K) > 0;
}
inline constexpr int tune_gemm_8p_min_tiles() { return 160; }

inline constexpr int tune_gemm_mid() { return 1; }       // 1: gemm_mid_kernel takes the few-tile bf16 shapes (round 3: 38.4 -> 32.7 us at 2560x2048x2048)
// Row split of an output of one to three rounds of 256x256 tiles plus a short remainder (the reference's batch: [2560, 8192] = 320
// tiles on 256 CUs, a second round at a quarter of the chip): rows [0, m1) = the whole rounds on the persistent kernel, the tail rows
// on the few-tile kernel -- 102.6 -> 78.3 us at 2560x8192x2048 (tools/probes/gemm_msplit.py).  0 = no split.
inline int gemm8p_row_split(int M, int N, int K, int ldx, int ldw, int ldy) {
    const int G = gemm8p_num_cu(), tm = cdiv(M, 256), tn = cdiv(N, 256);
    const int rounds = tm * tn / G, rem = tm * tn % G;
    // (round 5 tried the same split for ANY number of whole rounds plus a remainder of a third to a half of a round -- config 4 at
    // B = 64: [45056, 2048] = 5.5 rounds, the remainder as one round of the few-tile kernel: mmgl_gemm_nt 310.6 -> 307.8 ms per step,
    // step 282.0 -> 282.3 ms: nothing, removed.  A partial round costs what it costs at K = 2048: a tile is 45 us, and every way of
    // cutting the remainder finer -- K splits with fp32 partial tiles, 128x128 tiles at 0.68 of the big kernel's rate -- pays about
    // what it saves; bench.py prints `batch_sweep` instead of choosing batches around it)
    if (rounds < 1 || rounds > 3 || rem == 0 || rem > G / 4 || (rounds * G) % tn) return 0;
    const int m1 = rounds * G / tn * 256;
    return (m1 > 0 && m1 < M && gemm_mid_supported(M - m1, N, K, ldx, ldw, ldy)) ? m1 : 0;
}
// launch_gemm8p with that split applied (no K-split scratch: the callers below bring none or use it for other plans)
inline int launch_gemm8p_rows(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, const bf16* resid, const bf16* zmask,
                              int M, int N, int K, int act, float scale, hipStream_t st, float* part = nullptr, size_t part_bytes = 0) {
    const int m1 = gemm8p_row_split(M, N, K, ldx, ldw, ldy);
    if (!m1) return launch_gemm8p(X, ldx, W, ldw, Y, ldy, bias, resid, zmask, M, N, K, act, scale, st, part, part_bytes);
    int rc = launch_gemm8p(X, ldx, W, ldw, Y, ldy, bias, resid, zmask, m1, N, K, act, scale, st);
    if (rc) return rc;
    const size_t o = (size_t)m1 * ldy;
    return launch_gemm_mid(X + (size_t)m1 * ldx, ldx, W, ldw, Y + o, ldy, bias, resid ? resid + o : nullptr, zmask ? zmask + o : nullptr, M - m1, N, K,
                           act, scale, st);
}

// out[C,R] = transpose(f(in[R,C])) with f = (* scale) and optional ReLU mask from yact[R,C] (> 0);
// optional colsum[C] (+)= sum over R of f(in) (fp32 atomics are avoided: one block owns a full column strip).
// 64x64 tiles through LDS; grid.x = column strips, block loops over all row tiles of its strip.
template <typename T, bool MASK>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, const T* __restrict__ yact,
                                                        T* __restrict__ out, T* __restrict__ colsum, int R, int C,
                                                        int ldo, float scale, int accumulate) {
    __shared__ float tile[64][65];
    __shared__ float csum[4][64];
    const int c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float cs = 0.f;
    for (int r0 = blockIdx.y * 64; r0 < ldo; r0 += gridDim.y * 64) {
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {
            const int r = r0 + i, c = c0 + tx;
            float v = 0.f;
            if (r < R && c < C) {
                v = (float)in[(size_t)r * C + c] * scale;
                if (MASK && !((float)yact[(size_t)r * C + c] > 0.f)) v = 0.f;
            }
            tile[i][tx] = v;
            cs += v;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;
            if (c < C && r < ldo) out[(size_t)c * ldo + r] = (T)tile[tx][i];   // rows R..ldo-1 are zero padding
        }
        __syncthreads();
    }
    if (colsum && gridDim.y == 1) {
        csum[ty][tx] = cs;
        __syncthreads();
        if (ty == 0 && c0 + tx < C) {
            float s = csum[0][tx] + csum[1][tx] + csum[2][tx] + csum[3][tx];
            if (accumulate) s += (float)colsum[c0 + tx];
            colsum[c0 + tx] = (T)s;
        }
    }
}

// dyp = dy * scale * (y > 0)
template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ out,
                                                        size_t n, float scale) {
    typedef typename GT<T>::chunk_t V;
    constexpr int VN = GT<T>::VN;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / VN; i += (size_t)gridDim.x * 256) {
        const V d = ((const V*)dy)[i], a = ((const V*)y)[i];
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) o[j] = ((float)a[j] > 0.f) ? (T)((float)d[j] * scale) : (T)0.f;
        ((V*)out)[i] = o;
    }
}

inline constexpr bool tune_gemm_glds() { return true; }

template <typename T> inline int pad_k(int k) { return (k + GT<T>::VN - 1) / GT<T>::VN * GT<T>::VN; }

template <typename T>
int launch_gemm(const T* X, const T* W, T* Y, const T* bias, int M, int N, int K, int act, float scale, int accumulate,
                const T* X2, const T* W2, int K2, hipStream_t st, const T* zmask = nullptr, bool* zmask_done = nullptr) {
    constexpr int VN = GT<T>::VN;
    MMGL_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: M,N,K must be positive (got %d,%d,%d)", M, N, K);
    if (K % VN || N % 4 || (X2 && K2 % VN))
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm: K (%d) must be a multiple of %d and N (%d) of 4", K, VN, N);
    if constexpr (sizeof(T) == 2) {
        // persistent ping-pong kernel (gemm8p.hip) whenever the shape gives it enough 256x256 tiles
        if (!X2 && !accumulate && tune_gemm_8p() && gemm8p_supported(M, N, K, K, K, N) && cdiv(M, 256) * cdiv(N, 256) >= tune_gemm_8p_min_tiles()) {
            if (zmask_done) *zmask_done = zmask != nullptr;
            return launch_gemm8p_rows((const bf16*)X, K, (const bf16*)W, K, (bf16*)Y, N, (const bf16*)bias, nullptr, (const bf16*)zmask, M, N, K, act, scale, st);
        }
    }
    if constexpr (sizeof(T) == 2) {
        // 128x128 tiles, two workgroups per CU (gemm_mid.hip): outputs of too few 256x256 tiles for the persistent kernel
        if (!X2 && !accumulate && tune_gemm_mid() && gemm_mid_supported(M, N, K, K, K, N)) {
            if (zmask_done) *zmask_done = zmask != nullptr;
            return launch_gemm_mid((const bf16*)X, K, (const bf16*)W, K, (bf16*)Y, N, (const bf16*)bias, nullptr, (const bf16*)zmask, M, N, K, act, scale, st);
        }
    }
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    const size_t lds = 4 * TILE_BYTES;
    const bool glds = (K % GT<T>::BK == 0) && (!X2 || K2 % GT<T>::BK == 0) && tune_gemm_glds();
    const void* kern;
    if (glds) kern = act == MMGL_ACT_RELU ? (const void*)gemm_nt_kernel<T, MMGL_ACT_RELU, true> : (const void*)gemm_nt_kernel<T, MMGL_ACT_NONE, true>;
    else kern = act == MMGL_ACT_RELU ? (const void*)gemm_nt_kernel<T, MMGL_ACT_RELU, false> : (const void*)gemm_nt_kernel<T, MMGL_ACT_NONE, false>;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    dim3 grid(tiles_m * tiles_n), block(256);
#define NT_LAUNCH(A, G) hipLaunchKernelGGL((gemm_nt_kernel<T, A, G>), grid, block, lds, st, X, W, Y, bias, M, N, K, scale, accumulate, X2, W2, K2, tiles_m, tiles_n)
    if (glds) { if (act == MMGL_ACT_RELU) NT_LAUNCH(MMGL_ACT_RELU, true); else NT_LAUNCH(MMGL_ACT_NONE, true); }
    else { if (act == MMGL_ACT_RELU) NT_LAUNCH(MMGL_ACT_RELU, false); else NT_LAUNCH(MMGL_ACT_NONE, false); }
#undef NT_LAUNCH
    MMGL_CHECK_LAUNCH("gemm_nt");
    return MMGL_OK;
}

template <typename T>
int launch_transpose(const T* in, const T* yact, T* out, T* colsum, int R, int C, float scale, int accumulate, hipStream_t st) {
    const int ldo = pad_k<T>(R);   // the transposed matrix becomes a GEMM operand: its rows must be whole 16-B chunks
    dim3 grid(cdiv(C, 64), colsum ? 1 : (cdiv(ldo, 64) > 64 ? 64 : cdiv(ldo, 64)));
    if (yact) hipLaunchKernelGGL((transpose_kernel<T, true>), grid, dim3(256), 0, st, in, yact, out, colsum, R, C, ldo, scale, accumulate);
    else hipLaunchKernelGGL((transpose_kernel<T, false>), grid, dim3(256), 0, st, in, yact, out, colsum, R, C, ldo, scale, accumulate);
    MMGL_CHECK_LAUNCH("transpose");
    return MMGL_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Backward GEMMs without transposes (bf16).   out[RB][RA] = scale * sum_k  B(k, rb) * A(k, ra)   (+= if accumulate)
//   dgrad: out = dx[M][K_in],  A = W[N][K_in] stored k-major (TA), B = dy[M][N] stored row-major, contraction over N
//   wgrad: out = dW[N][K_in],  A = x[M][K_in] k-major (TA), B = dy[M][N] k-major (TB), contraction over M
// A k-major operand tile sits in LDS exactly as it is in memory ([64 k][128 cols], coalesced 256-B rows, row stride
// 288 B) and its MFMA fragments come out of ds_read_b64_tr_b16: per 16-lane group, lane i points at 4 consecutive
// columns of k-row 4g + (i>>2) and receives the 4 k-values of column i&15.  Two reads (k +0, +16) give the k order
// {4g..4g+3, 16+4g..}; a row-major partner operand is read in the same order with two ds_read_b64 (any k permutation
// is legal when both operands share it).  The ReLU mask of fc1 (dy * (y > 0)) is applied while staging.
constexpr int KROW = 288;                            // bytes per k-row of a k-major tile (256 + 32 pad: conflict-free tr reads)
constexpr int TX_TILE_BYTES = 64 * KROW;             // 18 KiB >= the 16 KiB of a row-major tile

__device__ __forceinline__ bf16x8 tfrag_kmajor(const char* tile, int blk, int ks, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int i = lane & 15, g = lane >> 4;
    const char* p = tile + (ks * 32 + 4 * g + (i >> 2)) * KROW + (blk * 16 + (i & 3) * 4) * 2;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * KROW));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
// row-major (swizzled) tile, same k order as tfrag_kmajor: k = 32 ks + {4g..4g+3, 16+4g..16+4g+3}
__device__ __forceinline__ bf16x8 rfrag_perm(const char* tile, int row, int ks, int g) {
    const bf16x4 lo = *(const bf16x4*)(tile + swz(row, ks * 4 + (g >> 1)) + (g & 1) * 8);
    const bf16x4 hi = *(const bf16x4*)(tile + swz(row, ks * 4 + 2 + (g >> 1)) + (g & 1) * 8);
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

// k-major tile for the LDS-DMA path: 256-B rows, no padding (the DMA destination is lane-linear), 32-B slot index XORed
// with (k-row & 7) on the SOURCE side so the 8 k-rows one tr16 cycle touches hit 8 distinct bank slots.
__device__ __forceinline__ bf16x8 tfrag_kmajor_swz(const char* tile, int blk, int ks, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int i = lane & 15, g = lane >> 4;
    const int row = ks * 32 + 4 * g + (i >> 2);             // row + 16 has the same (row & 7)
    const char* p = tile + row * 256 + ((blk ^ (row & 7)) << 5) + (i & 3) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 16 * 256));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

// The same fragment through INLINE-ASM transpose reads, for tiles filled by LDS-DMA: behind the builtin, hipcc's waitcnt pass
// cannot tell the read from the global_load_lds writes in flight and puts `s_waitcnt vmcnt(0)` in front of the first transpose
// read of every K tile -- the next tile's DMA, issued a few instructions earlier, was waited for before the current tile's
// MFMAs (no overlap at all).  The caller waits for lgkmcnt(0) itself before it uses the registers.
__device__ __forceinline__ bf16x8 tfrag_kmajor_swz_asm(const char* tile, int blk, int ks, int lane) {
    typedef __attribute__((address_space(3))) void lds_v;
    const int i = lane & 15, g = lane >> 4;
    const int row = ks * 32 + 4 * g + (i >> 2);
    const unsigned ad = (unsigned)(size_t)(lds_v*)tile + row * 256 + ((blk ^ (row & 7)) << 5) + (i & 3) * 8;
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(hi) : "v"(ad));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}

template <bool TB, bool MASK, bool GLDS>
__global__ __launch_bounds__(256) void gemm_tx_kernel(const bf16* __restrict__ Aop, int lda, const bf16* __restrict__ Bop,
                                                      int ldb, const bf16* __restrict__ Ymask, bf16* __restrict__ Out,
                                                      int RA, int RB, int K, float scale, int accumulate, int tiles_a,
                                                      int tiles_b) {
    typedef bf16x8 chunk_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                          // [2][TX_TILE_BYTES]  k-major
    char* sB = smem + 2 * TX_TILE_BYTES;      // [2][TX_TILE_BYTES]  k-major (TB) or row-major swizzled
    const int vid = xcd_remap(blockIdx.x, tiles_a * tiles_b);
    const int ta = vid / tiles_b, tb = vid % tiles_b;          // consecutive ids share the A panel
    const int a0 = ta * 128, b0 = tb * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = lane & 15, g = lane >> 4;
    const int wa = wave >> 1, wb = wave & 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = vzero<f32x4>();

    const int nk = (K + 63) / 64;
    chunk_t ra[4], rb[4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256;
            {   // A: k-major [64][128]
                const int row = id >> 4, c = id & 15;
                const int kk = k0 + row, col = a0 + c * 8;
                ra[i] = (kk < K && col < RA) ? *(const chunk_t*)(Aop + (size_t)kk * lda + col) : vzero<chunk_t>();
            }
            if constexpr (TB) {
                const int row = id >> 4, c = id & 15;
                const int kk = k0 + row, col = b0 + c * 8;
                const bool ok = kk < K && col < RB;
                chunk_t v = ok ? *(const chunk_t*)(Bop + (size_t)kk * ldb + col) : vzero<chunk_t>();
                if constexpr (MASK) {
                    if (ok) {
                        const chunk_t y = *(const chunk_t*)(Ymask + (size_t)kk * ldb + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = ((float)y[e] > 0.f) ? v[e] : (bf16)0.f;
                    }
                }
                rb[i] = v;
            } else {
                const int row = id >> 3, c = id & 7;
                const int kc = k0 + c * 8;
                const bool ok = b0 + row < RB && kc < K;
                chunk_t v = ok ? *(const chunk_t*)(Bop + (size_t)(b0 + row) * ldb + kc) : vzero<chunk_t>();
                if constexpr (MASK) {
                    if (ok) {
                        const chunk_t y = *(const chunk_t*)(Ymask + (size_t)(b0 + row) * ldb + kc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = ((float)y[e] > 0.f) ? v[e] : (bf16)0.f;
                    }
                }
                rb[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256;
            *(chunk_t*)(sA + buf * TX_TILE_BYTES + (id >> 4) * KROW + (id & 15) * 16) = ra[i];
            if constexpr (TB) *(chunk_t*)(sB + buf * TX_TILE_BYTES + (id >> 4) * KROW + (id & 15) * 16) = rb[i];
            else *(chunk_t*)(sB + buf * TX_TILE_BYTES + swz(id >> 3, id & 7)) = rb[i];
        }
    };

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto glds_kmajor = [&](const bf16* op, int ld, int col0, int k0, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = (i * 4 + wave_u) * 4;                        // 4 k-rows (1 KiB) per wave instruction
            const int row = rbase + (lane >> 4);
            const int s16 = lane & 15;
            const int c16 = ((((s16 >> 1) ^ (row & 7)) << 1) | (s16 & 1));
            const bf16* gp = op + (size_t)(k0 + row) * ld + col0 + c16 * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                             (__attribute__((address_space(3))) void*)(dst + rbase * 256), 16, 0, 0);
        }
    };
    auto glds_tile = [&](int kt, int buf) {
        const int k0 = kt * 64;
        glds_kmajor(Aop, lda, a0, k0, sA + buf * TX_TILE_BYTES);
        if constexpr (TB) glds_kmajor(Bop, ldb, b0, k0, sB + buf * TX_TILE_BYTES);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rbase = (i * 4 + wave_u) * 8;
                const int row = rbase + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                const bf16* gp = Bop + (size_t)min(b0 + row, RB - 1) * ldb + k0 + c * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(sB + buf * TX_TILE_BYTES + rbase * ROWB), 16, 0, 0);
            }
        }
    };

    if constexpr (GLDS) glds_tile(0, 0);
    else { load_tile(0); store_tile(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            if constexpr (GLDS) glds_tile(kt + 1, buf ^ 1);
            else load_tile(kt + 1);
        }
        const char* tA = sA + buf * TX_TILE_BYTES;
        const char* tB = sB + buf * TX_TILE_BYTES;
        if constexpr (GLDS) {
            // all 16 fragments of the K tile up front (inline-asm transpose reads: see tfrag_kmajor_swz_asm), one lgkmcnt(0), 32 MFMAs
            // under which the next tile's LDS-DMA -- issued above -- stays in flight; it is waited for just before the barrier
            bf16x8 fa[2][4], fb[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[ks][i] = tfrag_kmajor_swz_asm(tA, wa * 4 + i, ks, lane);
                    if constexpr (TB) fb[ks][i] = tfrag_kmajor_swz_asm(tB, wb * 4 + i, ks, lane);
                    else fb[ks][i] = rfrag_perm(tB, wb * 64 + i * 16 + x, ks, g);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(acc[i][j], fa[ks][i], fb[ks][j]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = tfrag_kmajor(tA, wa * 4 + i, ks, lane);
                    if constexpr (TB) fb[i] = tfrag_kmajor(tB, wb * 4 + i, ks, lane);
                    else fb[i] = rfrag_perm(tB, wb * 64 + i * 16 + x, ks, g);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mma16(acc[i][j], fa[i], fb[j]);
            }
            if (kt + 1 < nk) store_tile(buf ^ 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rbi = b0 + wb * 64 + j * 16 + x;
        if (rbi >= RB) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rai = a0 + wa * 64 + i * 16 + g * 4;
            if (rai >= RA) continue;
            f32x4 v = acc[i][j] * scale;
            bf16* op = Out + (size_t)rbi * RA + rai;
            if (accumulate) {
                const bf16x4 ov = *(const bf16x4*)op;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)ov[r];
            }
            *(bf16x4*)op = __builtin_convertvector(v, bf16x4);
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, bf16* __restrict__ out, size_t n4,
                                                            int nsplit, float scale, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = ((const f32x4*)part)[i];
        for (int s = 1; s < nsplit; ++s) a += ((const f32x4*)part)[(size_t)s * n4 + i];
        a *= scale;
        if (accumulate) {
            const bf16x4 o = ((const bf16x4*)out)[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += (float)o[r];
        }
        ((bf16x4*)out)[i] = __builtin_convertvector(a, bf16x4);
    }
}

inline size_t wgrad_partial_bytes(int RA, int RB, int K);
// the split-partial region of a linear's backward workspace: the weight gradient's K-split tiles (RA = in, RB = out features,
// contraction over the M rows) or, before them on the same stream, the dgrad's ([M, in] outputs, contraction over out features)
inline size_t split_partial_bytes(int RA, int RB, int K) {
    const size_t a = wgrad_partial_bytes(RA, RB, K), b = align_up(gemm8p_split_bytes(K, RA, RB), 256);
    return a > b ? a : b;
}
inline size_t wgrad_partial_bytes(int RA, int RB, int K) {
    const int s = gemm8p_tt_splits(RA, RB, K);
    return s > 1 ? align_up((size_t)s * RA * RB * sizeof(float), 256) : 0;
}

// column sums of dy (optionally ReLU-masked by y > 0): part[split][N] fp32, then colsum_finish folds the splits.
template <typename T, bool MASK>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, const T* __restrict__ y, float* __restrict__ part,
                                                     int M, int N) {
    typedef typename GT<T>::chunk_t V;
    constexpr int VN = GT<T>::VN;
    __shared__ float red[16][16 * 8 + 1];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int col = (blockIdx.x * 16 + cx) * VN;
    float a[VN];
#pragma unroll
    for (int e = 0; e < VN; ++e) a[e] = 0.f;
    if (col < N) {
        const int rows_per = (M + gridDim.y - 1) / gridDim.y;
        const int r0 = blockIdx.y * rows_per, r1 = min(r0 + rows_per, M);
        for (int r = r0 + ry; r < r1; r += 16) {
            const V d = *(const V*)(dy + (size_t)r * N + col);
            if constexpr (MASK) {
                const V yy = *(const V*)(y + (size_t)r * N + col);
#pragma unroll
                for (int e = 0; e < VN; ++e) a[e] += ((float)yy[e] > 0.f) ? (float)d[e] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < VN; ++e) a[e] += (float)d[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VN; ++e) red[ry][cx * VN + e] = a[e];
    __syncthreads();
    if (ry == 0 && col < N) {
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][cx * VN + e];
            part[(size_t)blockIdx.y * N + col + e] = t;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, T* __restrict__ out, int N, int splits,
                                                            float scale, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;        // four load streams per column instead of one serial chain of `splits` round trips
    int s = 0;
    for (; s + 3 < splits; s += 4) {
        t0 += part[(size_t)s * N + c];
        t1 += part[(size_t)(s + 1) * N + c];
        t2 += part[(size_t)(s + 2) * N + c];
        t3 += part[(size_t)(s + 3) * N + c];
    }
    for (; s < splits; ++s) t0 += part[(size_t)s * N + c];
    float t = (t0 + t1) + (t2 + t3);
    t *= scale;
    if (accumulate) t += (float)out[c];
    out[c] = (T)t;
}
constexpr int COLSUM_SPLITS = 64;       // upper bound (scratch sizing); a launch uses colsum_splits(N) of them
// about 1024 workgroups per launch: 16 splits of the rows left a 2048-column gradient on 256 workgroups, one per CU (57 us for 168 MB)
inline int colsum_splits(int M, int N, int vn) {
    const int cb = cdiv(N, 16 * vn);
    int s = (1024 + cb - 1) / cb;
    s = s < 16 ? 16 : (s > COLSUM_SPLITS ? COLSUM_SPLITS : s);
    const int by_rows = M / 128;                       // at least 8 row trips per workgroup (the reference's batch: M = 2560 -> 20 splits)
    if (s > by_rows) s = by_rows < 1 ? 1 : by_rows;
    return s;
}

template <typename T>
int launch_colsum(const T* dy, const T* y, T* out, float* part, int M, int N, float scale, int accumulate, hipStream_t st) {
    constexpr int VN = GT<T>::VN;
    if (N % VN) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "bias gradient: out_features %d must be a multiple of %d", N, VN);
    const int splits = colsum_splits(M, N, VN);
    dim3 grid(cdiv(N, 16 * VN), splits);
    if (y) hipLaunchKernelGGL((colsum_kernel<T, true>), grid, dim3(256), 0, st, dy, y, part, M, N);
    else hipLaunchKernelGGL((colsum_kernel<T, false>), grid, dim3(256), 0, st, dy, y, part, M, N);
    hipLaunchKernelGGL(colsum_finish_kernel<T>, dim3(cdiv(N, 256)), dim3(256), 0, st, part, out, N, splits, scale, accumulate);
    MMGL_CHECK_LAUNCH("colsum");
    return MMGL_OK;
}

int launch_gemm_tx(bool tb, const bf16* Aop, int lda, const bf16* Bop, int ldb, const bf16* ymask, bf16* Out, int RA, int RB,
                   int K, float scale, int accumulate, hipStream_t st, float* part = nullptr) {
    MMGL_CHECK_ARG(RA > 0 && RB > 0 && K > 0, "gemm_tx: bad sizes");
    if (RA % 8 || (tb ? RB % 8 : K % 8)) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "gemm_tx: feature dims must be multiples of 8");
    if (tb && !ymask && K >= 1024 && ((long long)K * lda * 2 >= 0xffffffffLL || (long long)K * ldb * 2 >= 0xffffffffLL)) {
        // a k-major operand of 4 GiB or more (the trainable lm_head's weight gradient at config 4 from B = 64 on: dlogits is
        // [45056, 50272] bf16 = 4.5 GB) does not fit one buffer descriptor: contract the two halves of the rows one after the other,
        // the second accumulating into the first one's output
        // (the caller sized `part` for the split count of the FULL contraction: a half whose own split count would need more
        // fp32 partial tiles than that runs without scratch -- unsplit, or on the 128x128 kernel)
        const int K1 = (K / 2 + 255) / 256 * 256;
        const size_t have = wgrad_partial_bytes(RA, RB, K);
        float* p1 = wgrad_partial_bytes(RA, RB, K1) <= have ? part : nullptr;
        float* p2 = wgrad_partial_bytes(RA, RB, K - K1) <= have ? part : nullptr;
        int rc = launch_gemm_tx(tb, Aop, lda, Bop, ldb, nullptr, Out, RA, RB, K1, scale, accumulate, st, p1);
        if (rc) return rc;
        return launch_gemm_tx(tb, Aop + (size_t)K1 * lda, lda, Bop + (size_t)K1 * ldb, ldb, nullptr, Out, RA, RB, K - K1, scale, 1, st, p2);
    }
    if (tb && !ymask && tune_gemm_8p()) {
        // both operands k-major and enough (tile, K split) work items: the ping-pong weight-gradient kernel (gemm8p_tt.hip)
        const int s8 = gemm8p_tt_splits(RA, RB, K);
        if (s8 == 1 || (s8 > 1 && part)) {
            int rc = launch_gemm8p_tt(Aop, lda, Bop, ldb, Out, part, RA, RB, K, s8, scale, accumulate, st);
            if (rc || s8 == 1) return rc;
            const size_t n4 = (size_t)RA * RB / 4;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256)), dim3(256), 0, st, part,
                               Out, n4, s8, scale, accumulate);
            MMGL_CHECK_LAUNCH("splitk_reduce");
            return MMGL_OK;
        }
    }
    const int tiles_a = cdiv(RA, 128), tiles_b = cdiv(RB, 128);
    const size_t lds = 4 * TX_TILE_BYTES;
    // LDS-DMA staging needs whole tiles along every k-major dimension and no ReLU mask (the mask is applied once, upstream)
    const bool glds = !ymask && K % 64 == 0 && RA % 128 == 0 && (!tb || RB % 128 == 0) && tune_gemm_glds();
    dim3 grid(tiles_a * tiles_b), block(256);
#define TX_LAUNCH(TBV, MK, GL)                                                                                              \
    do {                                                                                                                    \
        auto kf = gemm_tx_kernel<TBV, MK, GL>;                                                                              \
        hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        if (e != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));                      \
        hipLaunchKernelGGL(kf, grid, block, lds, st, Aop, lda, Bop, ldb, ymask, Out, RA, RB, K, scale, accumulate, tiles_a, tiles_b); \
    } while (0)
    if (glds) { if (tb) TX_LAUNCH(true, false, true); else TX_LAUNCH(false, false, true); }
    else if (tb) { if (ymask) TX_LAUNCH(true, true, false); else TX_LAUNCH(true, false, false); }
    else { if (ymask) TX_LAUNCH(false, true, false); else TX_LAUNCH(false, false, false); }
#undef TX_LAUNCH
    MMGL_CHECK_LAUNCH("gemm_tx");
    return MMGL_OK;
}

template <typename T>
int launch_relu_mask(const T* dy, const T* y, T* out, size_t n, float scale, hipStream_t st) {
    int blocks = (int)((n / GT<T>::VN + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(relu_mask_kernel<T>, dim3(blocks), dim3(256), 0, st, dy, y, out, n, scale);
    MMGL_CHECK_LAUNCH("relu_mask");
    return MMGL_OK;
}

// bf16 backward workspace: [dyp M*N (only with an activation)] [bias column-sum partials] [W^T K*N (big-tile dgrad)]
inline size_t bf16_wt_offset(int M, int N, int act) {
    return (act ? align_up((size_t)M * N * 2, 256) : 0) + align_up((size_t)COLSUM_SPLITS * N * sizeof(float), 256);
}
// ... [fp32 split-K partials of the weight gradient (small outputs only)]
inline size_t bf16_part_offset(int M, int N, int K, int act) { return bf16_wt_offset(M, N, act) + align_up((size_t)K * N * 2, 256); }

template <typename T>
int linear_dgrad(const T* dy, const T* y, const T* W, T* dx, char* ws, int M, int N, int K, int act, float scale, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {     // bf16: no transposes; dyp = dy*scale*(y>0) is materialised ONCE (masking inside the
        // GEMM would re-read y for every output tile column: 16x the L2 traffic at fc1's shape)
        const bf16* a = (const bf16*)dy;
        float sc = scale;
        if (act == MMGL_ACT_RELU) {
            int rc = launch_relu_mask<T>(dy, y, (T*)ws, (size_t)M * N, scale, st);
            if (rc) return rc;
            a = (const bf16*)ws;
            sc = 1.f;
        }
        if (big_tile_shape(M, K, N)) {
            T* Wt = (T*)(ws + bf16_wt_offset(M, N, act));
            int rc = launch_transpose<T>(W, nullptr, Wt, nullptr, N, K, 1.f, 0, st);
            if (rc) return rc;
            return launch_gemm<T>((const T*)a, Wt, dx, nullptr, M, K, N, MMGL_ACT_NONE, sc, 0, nullptr, nullptr, 0, st);
        }
        return launch_gemm_tx(false, (const bf16*)W, K, a, N, nullptr, (bf16*)dx, K, M, N, sc, 0, st);
    }
    // fp32 (parity path): W^T [K,N] | dyp [M,N] (only with an activation) in ws
    if (N % GT<T>::VN) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "linear_dgrad: out_features %d must be a multiple of %d", N, GT<T>::VN);
    T* Wt = (T*)ws;
    T* dyp = (T*)(ws + align_up((size_t)K * ((size_t)(N + 7) / 8 * 8) * sizeof(T), 256));
    int rc = launch_transpose<T>(W, nullptr, Wt, nullptr, N, K, 1.f, 0, st);
    if (rc) return rc;
    const T* a = dy;
    float s = scale;
    if (act == MMGL_ACT_RELU) {
        size_t n = (size_t)M * N;
        int blocks = (int)((n / GT<T>::VN + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(relu_mask_kernel<T>, dim3(blocks), dim3(256), 0, st, dy, y, dyp, n, scale);
        MMGL_CHECK_LAUNCH("relu_mask");
        a = dyp;
        s = 1.f;
    }
    return launch_gemm<T>(a, Wt, dx, nullptr, M, K, N, MMGL_ACT_NONE, s, 0, nullptr, nullptr, 0, st);   // N % 8 == 0 checked by caller
}

template <typename T>
int linear_wgrad(const T* dy, const T* y, const T* x, T* dW, T* dbias, char* ws, int M, int N, int K, int act, float scale,
                 int accumulate, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {     // bf16: both operands k-major straight from memory; ws = [dyp] | bias partials
        const bf16* a = (const bf16*)dy;
        float sc = scale;
        char* part = ws;
        if (act == MMGL_ACT_RELU) {
            int rc = launch_relu_mask<T>(dy, y, (T*)ws, (size_t)M * N, scale, st);
            if (rc) return rc;
            a = (const bf16*)ws;
            sc = 1.f;
            part = ws + align_up((size_t)M * N * sizeof(T), 256);
        }
        int rc = launch_gemm_tx(true, (const bf16*)x, K, a, N, nullptr, (bf16*)dW, K, N, M, sc, accumulate, st,
                                (float*)(ws + bf16_part_offset(M, N, K, act)));
        if (rc || !dbias) return rc;
        return launch_colsum<T>((const T*)a, nullptr, dbias, (float*)part, M, N, sc, accumulate, st);
    }
    // fp32 (parity path): dyp^T [N,M] | x^T [K,M] in ws
    T* dyT = (T*)ws;
    T* xT = (T*)(ws + align_up((size_t)N * ((size_t)(M + 7) / 8 * 8) * sizeof(T), 256));
    int rc = launch_transpose<T>(dy, act == MMGL_ACT_RELU ? y : nullptr, dyT, dbias, M, N, scale, accumulate, st);
    if (rc) return rc;
    rc = launch_transpose<T>(x, nullptr, xT, nullptr, M, K, 1.f, 0, st);
    if (rc) return rc;
    return launch_gemm<T>(dyT, xT, dW, nullptr, N, K, pad_k<T>(M), MMGL_ACT_NONE, 1.f, accumulate, nullptr, nullptr, 0, st);
}

size_t dgrad_ws(int M, int N, int K, int act, size_t esz) {
    const size_t Np = (size_t)(N + 7) / 8 * 8;
    if (esz == 2) return bf16_part_offset(M, N, K, act) + split_partial_bytes(K, N, M);
    return align_up((size_t)K * Np * esz, 256) + (act ? align_up((size_t)M * N * esz, 256) : 0);
}
size_t wgrad_ws(int M, int N, int K, size_t esz) {
    const size_t Mp = (size_t)(M + 7) / 8 * 8;
    if (esz == 2) return bf16_part_offset(M, N, K, 1) + split_partial_bytes(K, N, M);
    return align_up((size_t)N * Mp * esz, 256) + align_up((size_t)K * Mp * esz, 256);
}

template <typename T>
int lora_fwd(const T* x, const T* W, const T* bias, const T* A, const T* Bm, T* y, T* xa, int M, int N, int K, int r,
             float scale, hipStream_t st) {
    // xa = scale * x A^T ; y = [x | xa] . [W | Bm]^T + bias  (the LoRA term rides in the same accumulators)
    int rc = launch_gemm<T>(x, A, xa, nullptr, M, r, K, MMGL_ACT_NONE, scale, 0, nullptr, nullptr, 0, st);
    if (rc) return rc;
    return launch_gemm<T>(x, W, y, bias, M, N, K, MMGL_ACT_NONE, 1.f, 0, xa, Bm, r, st);
}

template <typename T>
int lora_bwd(const T* dy, const T* x, const T* xa, const T* W, const T* A, const T* Bm, T* dx, T* dA, T* dB, T* dyb,
             char* ws, int M, int N, int K, int r, float scale, int accumulate, hipStream_t st) {
    // ws: W^T [K,N] | A^T [K,r] | Bm^T [r,N] | dy^T [N,M] | x^T [K,M] | xa^T [r,M] | dyb^T [r,M]
    if (N % GT<T>::VN || r % GT<T>::VN)
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "lora_bwd: out_features %d and rank %d must be multiples of %d", N, r, GT<T>::VN);
    size_t off = 0;
    auto take = [&](size_t elems) { T* p = (T*)(ws + off); off += align_up(elems * sizeof(T), 256); return p; };
    const size_t Np = (size_t)(N + 7) / 8 * 8, rp = (size_t)(r + 7) / 8 * 8;
    T* Wt = take((size_t)K * Np); T* At = take((size_t)K * rp); T* Bt = take((size_t)r * Np);
    const size_t Mp = (size_t)(M + 7) / 8 * 8;
    T* dyT = take((size_t)N * Mp); T* xT = take((size_t)K * Mp); T* xaT = take((size_t)r * Mp); T* dybT = take((size_t)r * Mp);
    int rc;
    if ((rc = launch_transpose<T>(W, nullptr, Wt, nullptr, N, K, 1.f, 0, st))) return rc;
    if ((rc = launch_transpose<T>(A, nullptr, At, nullptr, r, K, 1.f, 0, st))) return rc;
    if ((rc = launch_transpose<T>(Bm, nullptr, Bt, nullptr, N, r, 1.f, 0, st))) return rc;
    // dyb = scale * dy Bm            [M,r]
    if ((rc = launch_gemm<T>(dy, Bt, dyb, nullptr, M, r, N, MMGL_ACT_NONE, scale, 0, nullptr, nullptr, 0, st))) return rc;
    // dx = dy W + dyb A              [M,K]
    if ((rc = launch_gemm<T>(dy, Wt, dx, nullptr, M, K, N, MMGL_ACT_NONE, 1.f, 0, dyb, At, r, st))) return rc;
    // dA = dyb^T x  [r,K] ; dB = dy^T xa [N,r]   (xa already carries `scale`)
    if ((rc = launch_transpose<T>(dyb, nullptr, dybT, nullptr, M, r, 1.f, 0, st))) return rc;
    if ((rc = launch_transpose<T>(x, nullptr, xT, nullptr, M, K, 1.f, 0, st))) return rc;
    if ((rc = launch_gemm<T>(dybT, xT, dA, nullptr, r, K, pad_k<T>(M), MMGL_ACT_NONE, 1.f, accumulate, nullptr, nullptr, 0, st))) return rc;
    if ((rc = launch_transpose<T>(dy, nullptr, dyT, nullptr, M, N, 1.f, 0, st))) return rc;
    if ((rc = launch_transpose<T>(xa, nullptr, xaT, nullptr, M, r, 1.f, 0, st))) return rc;
    return launch_gemm<T>(dyT, xaT, dB, nullptr, N, r, pad_k<T>(M), MMGL_ACT_NONE, 1.f, accumulate, nullptr, nullptr, 0, st);
}

size_t lora_ws(int M, int N, int K, int r, size_t esz) {
    size_t t = 0;
    const size_t Mp = (size_t)(M + 7) / 8 * 8, Np = (size_t)(N + 7) / 8 * 8, rp = (size_t)(r + 7) / 8 * 8;
    size_t e[7] = {(size_t)K * Np, (size_t)K * rp, (size_t)r * Np, (size_t)N * Mp, (size_t)K * Mp, (size_t)r * Mp, (size_t)r * Mp};
    for (int i = 0; i < 7; ++i) t += align_up(e[i] * esz, 256);
    return t;
}

}  // namespace

#define DT_SWITCH(who, EXPR_BF16, EXPR_F32)                               \
    if (dtype == MMGL_BF16) return EXPR_BF16;                             \
    if (dtype == MMGL_F32) return EXPR_F32;                               \
    MMGL_FAIL(MMGL_ERR_INVALID, "%s: bad dtype %d", who, dtype)

extern "C" int mmgl_linear_fwd(const void* x, const void* W, const void* bias, void* y, int M, int N, int K, int act,
                               float out_scale, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && W && y, "mmgl_linear_fwd: null pointer");
    MMGL_CHECK_ARG(act == MMGL_ACT_NONE || act == MMGL_ACT_RELU, "mmgl_linear_fwd: unknown activation %d", act);
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("mmgl_linear_fwd",
              launch_gemm<bf16>((const bf16*)x, (const bf16*)W, (bf16*)y, (const bf16*)bias, M, N, K, act, out_scale, 0, nullptr, nullptr, 0, st),
              launch_gemm<float>((const float*)x, (const float*)W, (float*)y, (const float*)bias, M, N, K, act, out_scale, 0, nullptr, nullptr, 0, st));
}

extern "C" size_t mmgl_linear_dgrad_workspace(int M, int N, int K, int act, int dtype) {
    return dgrad_ws(M, N, K, act, dtype == MMGL_BF16 ? 2 : 4);
}
extern "C" size_t mmgl_linear_wgrad_workspace(int M, int N, int K, int dtype) { return wgrad_ws(M, N, K, dtype == MMGL_BF16 ? 2 : 4); }

extern "C" int mmgl_linear_dgrad(const void* dy, const void* y, const void* W, void* dx, void* workspace, size_t workspace_bytes,
                                 int M, int N, int K, int act, float out_scale, int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && W && dx && workspace && (act == MMGL_ACT_NONE || y), "mmgl_linear_dgrad: null pointer");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_linear_dgrad_workspace(M, N, K, act, dtype), "mmgl_linear_dgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    DT_SWITCH("mmgl_linear_dgrad",
              linear_dgrad<bf16>((const bf16*)dy, (const bf16*)y, (const bf16*)W, (bf16*)dx, ws, M, N, K, act, out_scale, st),
              linear_dgrad<float>((const float*)dy, (const float*)y, (const float*)W, (float*)dx, ws, M, N, K, act, out_scale, st));
}

extern "C" int mmgl_linear_wgrad(const void* dy, const void* y, const void* x, void* dW, void* dbias, void* workspace,
                                 size_t workspace_bytes, int M, int N, int K, int act, float out_scale, int accumulate,
                                 int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && x && dW && workspace && (act == MMGL_ACT_NONE || y), "mmgl_linear_wgrad: null pointer");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_linear_wgrad_workspace(M, N, K, dtype), "mmgl_linear_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    DT_SWITCH("mmgl_linear_wgrad",
              linear_wgrad<bf16>((const bf16*)dy, (const bf16*)y, (const bf16*)x, (bf16*)dW, (bf16*)dbias, ws, M, N, K, act, out_scale, accumulate, st),
              linear_wgrad<float>((const float*)dy, (const float*)y, (const float*)x, (float*)dW, (float*)dbias, ws, M, N, K, act, out_scale, accumulate, st));
}

// One call for the whole backward of a linear: dyp once, then dx / dW / dbias as requested (any of them may be NULL).
extern "C" int mmgl_gemm_nt_fast(int M, int N, int K, int ldx, int ldw, int ldy, int dtype);

template <typename T>
int linear_bwd(const T* dy, const T* y, const T* x, const T* W, T* dx, T* dW, T* dbias, char* ws, int M, int N, int K, int act,
               float scale, int accumulate, hipStream_t st, bool mask_dx = false) {
    // mask_dx: x is the output of a ReLU; its backward (dx = 0 where x <= 0) is folded into this dgrad -- the epilogue of the
    // 256x256 kernel, or one in-place pass after the other GEMM forms
    if constexpr (sizeof(T) == 2) {
        const bf16* a = (const bf16*)dy;
        float sc = scale;
        char* part = ws;
        if (act == MMGL_ACT_RELU) {
            int rc = launch_relu_mask<T>(dy, y, (T*)ws, (size_t)M * N, scale, st);
            if (rc) return rc;
            a = (const bf16*)ws;
            sc = 1.f;
            part = ws + align_up((size_t)M * N * sizeof(T), 256);
        }
        int rc = MMGL_OK;
        if (dx) {
            if ((big_tile_shape(M, K, N) || (N % 128 == 0 && mmgl_gemm_nt_fast(M, K, N, N, N, K, MMGL_BF16)))) {
                // (sending every M >= 1024 dgrad through W^T and the 128x128 NT kernel was tried: no gain in the batch-4 step)
                // shapes the persistent kernel takes (a chip of 256x256 tiles, or K-split work items at the reference's small
                // batch): dx = dyp . (W^T)^T as an NT GEMM; transposing the [N,K] weight costs a few percent of the GEMM
                T* Wt = (T*)(ws + bf16_wt_offset(M, N, act));
                rc = launch_transpose<T>(W, nullptr, Wt, nullptr, N, K, 1.f, 0, st);
                bool masked = false;
                if (!rc && tune_gemm_8p() && gemm8p_supported(M, K, N, N, N, K) && gemm8p_use_splits(M, K, N)) {
                    // few tiles, or a remainder of a round of tiles (the reference's batch): K-split work items, scratch tiles in
                    // the region the weight gradient's split partials use afterwards (same stream: the dgrad has consumed them by then)
                    rc = launch_gemm8p_rows((const bf16*)a, N, (const bf16*)Wt, N, (bf16*)dx, K, nullptr, nullptr, mask_dx ? (const bf16*)x : nullptr,
                                       M, K, N, MMGL_ACT_NONE, sc, st, (float*)(ws + bf16_part_offset(M, N, K, act)), gemm8p_split_bytes(M, K, N));
                    masked = mask_dx;
                } else if (!rc)
                    rc = launch_gemm<T>((const T*)a, Wt, dx, nullptr, M, K, N, MMGL_ACT_NONE, sc, 0, nullptr, nullptr, 0, st,
                                        mask_dx ? x : nullptr, &masked);
                if (!rc && mask_dx && !masked) rc = launch_relu_mask<T>(dx, x, dx, (size_t)M * K, 1.f, st);
            } else {
                rc = launch_gemm_tx(false, (const bf16*)W, K, a, N, nullptr, (bf16*)dx, K, M, N, sc, 0, st);
                if (!rc && mask_dx) rc = launch_relu_mask<T>(dx, x, dx, (size_t)M * K, 1.f, st);
            }
        }
        if (!rc && dW) rc = launch_gemm_tx(true, (const bf16*)x, K, a, N, nullptr, (bf16*)dW, K, N, M, sc, accumulate, st,
                                           (float*)(ws + bf16_part_offset(M, N, K, act)));
        if (!rc && dbias) rc = launch_colsum<T>((const T*)a, nullptr, dbias, (float*)part, M, N, sc, accumulate, st);
        return rc;
    } else {
        int rc = MMGL_OK;
        if (dx) rc = linear_dgrad<T>(dy, y, W, dx, ws, M, N, K, act, scale, st);
        if (!rc && dx && mask_dx) rc = launch_relu_mask<T>(dx, x, dx, (size_t)M * K, 1.f, st);
        if (!rc && (dW || dbias)) {
            MMGL_CHECK_ARG(dW, "mmgl_linear_bwd: fp32 path needs dW when dbias is requested");
            rc = linear_wgrad<T>(dy, y, x, dW, dbias, ws, M, N, K, act, scale, accumulate, st);
        }
        return rc;
    }
}

extern "C" size_t mmgl_linear_bwd_workspace(int M, int N, int K, int act, int dtype) {
    const size_t esz = dtype == MMGL_BF16 ? 2 : 4;
    const size_t a = dgrad_ws(M, N, K, act, esz), b = wgrad_ws(M, N, K, esz);
    return a > b ? a : b;
}

extern "C" int mmgl_linear_bwd(const void* dy, const void* y, const void* x, const void* W, void* dx, void* dW, void* dbias,
                               void* workspace, size_t workspace_bytes, int M, int N, int K, int act, float out_scale,
                               int accumulate, int mask_dx, int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && x && W && workspace && (act == MMGL_ACT_NONE || y), "mmgl_linear_bwd: null pointer");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_linear_bwd_workspace(M, N, K, act, dtype), "mmgl_linear_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    DT_SWITCH("mmgl_linear_bwd",
              linear_bwd<bf16>((const bf16*)dy, (const bf16*)y, (const bf16*)x, (const bf16*)W, (bf16*)dx, (bf16*)dW, (bf16*)dbias, ws, M, N, K, act, out_scale, accumulate, st, mask_dx != 0),
              linear_bwd<float>((const float*)dy, (const float*)y, (const float*)x, (const float*)W, (float*)dx, (float*)dW, (float*)dbias, ws, M, N, K, act, out_scale, accumulate, st, mask_dx != 0));
}

extern "C" int mmgl_lora_linear_fwd(const void* x, const void* W, const void* bias, const void* A, const void* Bm, void* y,
                                    void* xa, int M, int N, int K, int r, float scale, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && W && A && Bm && y && xa && r > 0, "mmgl_lora_linear_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("mmgl_lora_linear_fwd",
              lora_fwd<bf16>((const bf16*)x, (const bf16*)W, (const bf16*)bias, (const bf16*)A, (const bf16*)Bm, (bf16*)y, (bf16*)xa, M, N, K, r, scale, st),
              lora_fwd<float>((const float*)x, (const float*)W, (const float*)bias, (const float*)A, (const float*)Bm, (float*)y, (float*)xa, M, N, K, r, scale, st));
}

extern "C" size_t mmgl_lora_linear_bwd_workspace(int M, int N, int K, int r, int dtype) {
    return lora_ws(M, N, K, r, dtype == MMGL_BF16 ? 2 : 4);
}

extern "C" int mmgl_lora_linear_bwd(const void* dy, const void* x, const void* xa, const void* W, const void* A,
                                    const void* Bm, void* dx, void* dA, void* dB, void* dyb, void* workspace,
                                    size_t workspace_bytes, int M, int N, int K, int r, float scale, int accumulate,
                                    int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && x && xa && W && A && Bm && dx && dA && dB && dyb && workspace && r > 0, "mmgl_lora_linear_bwd: bad arguments");
    MMGL_CHECK_ARG(workspace_bytes >= mmgl_lora_linear_bwd_workspace(M, N, K, r, dtype), "mmgl_lora_linear_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    DT_SWITCH("mmgl_lora_linear_bwd",
              lora_bwd<bf16>((const bf16*)dy, (const bf16*)x, (const bf16*)xa, (const bf16*)W, (const bf16*)A, (const bf16*)Bm, (bf16*)dx, (bf16*)dA, (bf16*)dB, (bf16*)dyb, ws, M, N, K, r, scale, accumulate, st),
              lora_bwd<float>((const float*)dy, (const float*)x, (const float*)xa, (const float*)W, (const float*)A, (const float*)Bm, (float*)dx, (float*)dA, (float*)dB, (float*)dyb, ws, M, N, K, r, scale, accumulate, st));
}

extern "C" int mmgl_transpose(const void* in, void* out, int R, int C, int dtype, void* stream) {
    MMGL_CHECK_ARG(in && out && R > 0 && C > 0, "mmgl_transpose: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("mmgl_transpose", launch_transpose<bf16>((const bf16*)in, nullptr, (bf16*)out, nullptr, R, C, 1.f, 0, st),
              launch_transpose<float>((const float*)in, nullptr, (float*)out, nullptr, R, C, 1.f, 0, st));
}

// ------------------------------------------------------------------------------------------------------------------
// General NT GEMM with a fused epilogue (the frozen path's linears and their dgrads): fast path = gemm8p.hip; every other
// shape / dtype is composed from the kernels above plus the elementwise entry points.
extern "C" int mmgl_activation_fwd(const void* x, void* y, size_t n, int act, int dtype, void* stream);
extern "C" int mmgl_gated_residual_fwd(const void* residual, const void* x, const float* gate, void* y, size_t n, float p_drop,
                                       uint64_t seed, int dtype, void* stream);

static bool gemm_nt_on_8p(int M, int N, int K, int ldx, int ldw, int ldy, int dtype) {
    return dtype == MMGL_BF16 && tune_gemm_8p() && gemm8p_supported(M, N, K, ldx, ldw, ldy) &&
           (cdiv(M, 256) * cdiv(N, 256) >= tune_gemm_8p_min_tiles() || gemm8p_use_splits(M, N, K));
}
// 1: the persistent 256x256 kernel, 2: the 128x128 kernel (both: strided operands, whole epilogue in the kernel), 0: composed path
extern "C" int mmgl_gemm_nt_fast(int M, int N, int K, int ldx, int ldw, int ldy, int dtype) {
    if (gemm_nt_on_8p(M, N, K, ldx, ldw, ldy, dtype)) return 1;
    return (dtype == MMGL_BF16 && tune_gemm_mid() && gemm_mid_supported(M, N, K, ldx, ldw, ldy)) ? 2 : 0;
}

extern "C" size_t mmgl_gemm_nt_workspace(int M, int N, int K, int ldx, int ldw, int ldy, int dtype) {
    if (dtype != MMGL_BF16 || !tune_gemm_8p() || !gemm8p_supported(M, N, K, ldx, ldw, ldy)) return 0;
    if (!gemm8p_use_splits(M, N, K)) return 0;              // few-tile outputs and remainders of a round of tiles (gemm8p_plan)
    return align_up(gemm8p_split_bytes(M, N, K), 256);
}

extern "C" int mmgl_gemm_nt(const void* x, int ldx, const void* W, int ldw, const void* bias, const void* residual, const void* zmask,
                            void* y, int ldy, int M, int N, int K, int act, float out_scale, void* workspace, size_t workspace_bytes,
                            int dtype, void* stream) {
    MMGL_CHECK_ARG(x && W && y, "mmgl_gemm_nt: null pointer");
    MMGL_CHECK_ARG(act >= 0 && act <= 4, "mmgl_gemm_nt: unknown activation %d", act);
    // K may exceed x's row length by less than one 128-wide K step when the matching columns of W are zero padding (the
    // contraction over a vocabulary that is no multiple of 128): the tail of a row then reads the head of the next one
    MMGL_CHECK_ARG(ldx + 127 >= K && ldw >= K && ldy >= N, "mmgl_gemm_nt: leading dimensions (%d, %d, %d) smaller than the rows (K=%d, N=%d)", ldx, ldw, ldy, K, N);
    hipStream_t st = (hipStream_t)stream;
    // few-tile shapes run as K-split work items when the caller brought mmgl_gemm_nt_workspace() bytes, unsplit otherwise
    if (gemm_nt_on_8p(M, N, K, ldx, ldw, ldy, dtype))
        return launch_gemm8p_rows((const bf16*)x, ldx, (const bf16*)W, ldw, (bf16*)y, ldy, (const bf16*)bias, (const bf16*)residual,
                                  (const bf16*)zmask, M, N, K, act, out_scale, st, (float*)workspace, workspace_bytes);
    if (dtype == MMGL_BF16 && tune_gemm_mid() && gemm_mid_supported(M, N, K, ldx, ldw, ldy))
        return launch_gemm_mid((const bf16*)x, ldx, (const bf16*)W, ldw, (bf16*)y, ldy, (const bf16*)bias, (const bf16*)residual,
                               (const bf16*)zmask, M, N, K, act, out_scale, st);
    if (ldx != K || ldw != K || ldy != N)
        MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "mmgl_gemm_nt: strided operands (ld %d %d %d) need the bf16 fast path (K %% 128 == 0, N %% 16 == 0, "
                  ">= %d tiles of 256x256); got M=%d N=%d K=%d dtype=%d", ldx, ldw, ldy, tune_gemm_8p_min_tiles(), M, N, K, dtype);
    int rc = mmgl_linear_fwd(x, W, bias, y, M, N, K, act <= 1 ? act : 0, out_scale, dtype, stream);
    if (rc) return rc;
    const size_t n = (size_t)M * N;
    if (act >= 2 && (rc = mmgl_activation_fwd(y, y, n, act, dtype, stream))) return rc;
    if (zmask) {
        if (dtype == MMGL_BF16) rc = launch_relu_mask<bf16>((const bf16*)y, (const bf16*)zmask, (bf16*)y, n, 1.f, st);
        else rc = launch_relu_mask<float>((const float*)y, (const float*)zmask, (float*)y, n, 1.f, st);
        if (rc) return rc;
    }
    if (residual) rc = mmgl_gated_residual_fwd(residual, y, nullptr, y, n, 0.f, 0, dtype, stream);
    return rc;
}

// ReLU mask as bits.  fc1 of a frozen FFN writes one bit per element of relu(x W1^T + b1) beside the activation; fc2's dgrad
// (dh = (dy W2) where h > 0) applies them in its epilogue instead of re-reading the [M, ffn] activation: 16 bytes per lane and
// tile instead of 256, and the activation need not be kept for the backward pass.  Only for shapes that run as whole 256x256
// tiles on the persistent kernel (the bits are lane-private: both kernels must map tiles to lanes the same way).
extern "C" size_t mmgl_gemm_nt_relu_bits_bytes(int M, int N, int K, int ldx, int ldw, int ldy, int dtype) {
    if (dtype != MMGL_BF16 || !tune_gemm_8p() || !gemm8p_supported(M, N, K, ldx, ldw, ldy)) return 0;
    if (cdiv(M, 256) * cdiv(N, 256) < tune_gemm_8p_min_tiles()) return 0;
    return gemm8p_bits_bytes(M, N);
}

extern "C" int mmgl_gemm_nt_relu_bits(const void* x, int ldx, const void* W, int ldw, const void* bias, void* y, int ldy, void* bits_out,
                                      int M, int N, int K, float out_scale, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && W && y && bits_out, "mmgl_gemm_nt_relu_bits: null pointer");
    MMGL_CHECK_ARG(mmgl_gemm_nt_relu_bits_bytes(M, N, K, ldx, ldw, ldy, dtype) > 0,
                   "mmgl_gemm_nt_relu_bits: shape M=%d N=%d K=%d (ld %d %d %d, dtype %d) does not run as whole tiles of the persistent kernel", M, N, K, ldx, ldw, ldy, dtype);
    return launch_gemm8p((const bf16*)x, ldx, (const bf16*)W, ldw, (bf16*)y, ldy, (const bf16*)bias, nullptr, nullptr, M, N, K, MMGL_ACT_RELU,
                         out_scale, (hipStream_t)stream, nullptr, 0, (unsigned*)bits_out, nullptr);
}

extern "C" int mmgl_gemm_nt_masked(const void* x, int ldx, const void* W, int ldw, const void* bits_in, void* y, int ldy, int M, int N, int K,
                                   float out_scale, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && W && y && bits_in, "mmgl_gemm_nt_masked: null pointer");
    MMGL_CHECK_ARG(ldx + 127 >= K && ldw >= K && ldy >= N, "mmgl_gemm_nt_masked: leading dimensions (%d, %d, %d) smaller than the rows (K=%d, N=%d)", ldx, ldw, ldy, K, N);
    MMGL_CHECK_ARG(mmgl_gemm_nt_relu_bits_bytes(M, N, K, ldx, ldw, ldy, dtype) > 0,
                   "mmgl_gemm_nt_masked: shape M=%d N=%d K=%d (ld %d %d %d, dtype %d) does not run as whole tiles of the persistent kernel", M, N, K, ldx, ldw, ldy, dtype);
    return launch_gemm8p((const bf16*)x, ldx, (const bf16*)W, ldw, (bf16*)y, ldy, nullptr, nullptr, nullptr, M, N, K, MMGL_ACT_NONE, out_scale,
                         (hipStream_t)stream, nullptr, 0, nullptr, (const unsigned*)bits_in);
}

/* dy * (y > 0): backward of a stand-alone ReLU (in place allowed) */
extern "C" int mmgl_relu_bwd(const void* dy, const void* y, void* out, size_t n, int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && y && out, "mmgl_relu_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    DT_SWITCH("mmgl_relu_bwd", launch_relu_mask<bf16>((const bf16*)dy, (const bf16*)y, (bf16*)out, n, 1.f, st),
              launch_relu_mask<float>((const float*)dy, (const float*)y, (float*)out, n, 1.f, st));
}
