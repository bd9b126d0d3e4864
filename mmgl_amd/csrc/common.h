// Shared device/host helpers for libmmgl_hip.so (gfx950 / CDNA4 only: wave64, MFMA 16x16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/mmgl_hip.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------- host side
void mmgl_set_error(const char* fmt, ...);
#define MMGL_FAIL(code, ...)            \
    do {                                \
        mmgl_set_error(__VA_ARGS__);    \
        return (code);                  \
    } while (0)
#define MMGL_CHECK_ARG(cond, ...) \
    do {                          \
        if (!(cond)) MMGL_FAIL(MMGL_ERR_INVALID, __VA_ARGS__); \
    } while (0)
#define MMGL_CHECK_LAUNCH(name)                                                         \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) MMGL_FAIL(MMGL_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------------------------- device side
#define WAVE 64

template <typename T> struct Elem;
template <> struct Elem<float> {
    typedef f32x8 v8;
    typedef f32x4 v4;
    static __device__ __forceinline__ float to_f(float x) { return x; }
    static __device__ __forceinline__ float from_f(float x) { return x; }
};
template <> struct Elem<bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
    static __device__ __forceinline__ bf16 from_f(float x) { return (bf16)x; }
};

template <typename V> __device__ __forceinline__ V vzero() {
    V z;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(V) / sizeof(z[0])); ++i) z[i] = 0;
    return z;
}

// One 16x16 output tile, contraction over this lane-group's 8 k-slots.
//   acc[r] on lane (x = l&15, g = l>>4) is C[row = 4g + r][col = x]
//   a: lane (i, g) holds A[i][k(g, 0..7)]   b: lane (j, g) holds B[k(g, 0..7)][j]
// Any k(g,e) bijection is valid as long as A and B use the same one, which the kernels exploit to
// feed accumulator-layout data straight back in as an operand (no cross-lane shuffles).
__device__ __forceinline__ void mma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ typename Elem<T>::v8 pack8(const f32x4& lo, const f32x4& hi);
template <> __device__ __forceinline__ f32x8 pack8<float>(const f32x4& lo, const f32x4& hi) {
    f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
}
template <> __device__ __forceinline__ bf16x8 pack8<bf16>(const f32x4& lo, const f32x4& hi) {
    f32x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_convertvector(r, bf16x8);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware remap of a linear block id: consecutive virtual ids land on the same XCD (block b is
// observed to run on XCD b % 8) so work items that share operands share an L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, j = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

// GEMM tile order: virtual id -> (tm, tn) in bands of GN tile columns walked in groups of GM tile rows, so the GM*GN = 32
// workgroups an XCD runs at a time cover a GM x GN block: every K slice of an activation tile is fetched into that XCD's L2 once
// and hit GN-1 times, every K slice of a weight tile hit GM-1 times (the plain column-major order streams the whole activation
// matrix once per tile COLUMN).  Bijective for any tiles_m, tiles_n.
__device__ __forceinline__ void grouped_tile(int vid, int tiles_m, int tiles_n, int& tm, int& tn) {
    // 4 tile rows x 8 tile columns since the output tiles are stored non-temporally (8 x 4 before): the activation K slices, the
    // big operand at N = 2048, are hit 7 times per fetch; seven GEMM shapes of the step 3494 -> 3418 us, step +0.8 %, config 4 +1.7 %
    // (16 x 2: +2 %, 2 x 16: -1 %, 32 x 1: +8 %)
#ifndef MMGL_TILE_GM
#define MMGL_TILE_GM 4
#define MMGL_TILE_GN 8
#endif
    constexpr int GM = MMGL_TILE_GM, GN = MMGL_TILE_GN;
    int band = vid / (tiles_m * GN);
    const int last_band = (tiles_n - 1) / GN;
    band = band > last_band ? last_band : band;
    const int rem = vid - band * tiles_m * GN;
    const int ncols = min(GN, tiles_n - band * GN);
    int grp = rem / (GM * ncols);
    const int last_grp = (tiles_m - 1) / GM;
    grp = grp > last_grp ? last_grp : grp;
    const int r2 = rem - grp * GM * ncols;
    const int nrows = min(GM, tiles_m - grp * GM);
    tm = grp * GM + r2 % nrows;
    tn = band * GN + r2 / nrows;
}

// Counter-based dropout hash: keep-mask for element i under (seed).  splitmix64 finaliser.
__device__ __forceinline__ uint32_t mmgl_hash32(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
