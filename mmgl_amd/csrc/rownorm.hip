// LayerNorm / RMSNorm forward + backward over the last dim (HBM-bound row kernels, gfx950).
// replaces nn.LayerNorm at reference model/modelling_cross_attention.py:319-320, 340-341, 349-350, 364-365, 635-636.
//
// One row is owned by TPR threads (64 = one wave for cols <= 1024, else 256 = the whole block); every thread
// keeps its 16-byte chunks of the row in registers, so x is read from HBM exactly once (two-pass mean /
// variance in fp32, like torch).  Backward walks rows grid-stride so each thread can carry the dgamma/dbeta
// column partials of its own columns in registers; per-block partials go to the workspace and are summed in
// a fixed order by a second kernel (deterministic).
#include "common.h"

namespace {

constexpr int NVMAX = 4;          // 16-B chunks per thread: cols <= 256 * NVMAX * VEC (8192 bf16 / 4096 f32 -> see dispatch)
constexpr int MAXBLK = 512;       // backward grid cap: 2 blocks per CU; each block carries its column partials in registers

template <typename T> struct Vec {
    static constexpr int N = 16 / sizeof(T);
    typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
};

template <int TPR> __device__ __forceinline__ float row_sum(float v, float* red, int row_in_blk) {
    v = wave_sum(v);
    if constexpr (TPR == 64) return v;
    else {
        const int w = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[w] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    }
}

template <typename T, int TPR, int NV, bool RMS>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                       const T* __restrict__ gamma, const T* __restrict__ beta,
                                                       T* __restrict__ sum_out, T* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                       int cols, float eps, float p, uint64_t seed) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    __shared__ float red[4];
    const int rpb = 256 / TPR;
    const int tr = threadIdx.x % TPR, rib = threadIdx.x / TPR;
    const int nchunks = cols / VN;
    const float keep_scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    for (int row0 = blockIdx.x * rpb; row0 < rows; row0 += gridDim.x * rpb) {
        const int row = row0 + rib;
        const bool live = row < rows;
        float xv[NV][VN];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tr + i * TPR;
            if (live && c < nchunks) {
                V v = *(const V*)(x + (size_t)row * cols + c * VN);
                if (res) {                                    // fused residual add: the sum is rounded to T like torch's x + r
                    const V r = *(const V*)(res + (size_t)row * cols + c * VN);
                    if (p > 0.f) {                            // s = res + dropout(x): same counter hash as mmgl_gated_residual_fwd
                        const uint64_t e0 = (uint64_t)row * cols + (uint64_t)c * VN;
#pragma unroll
                        for (int j = 0; j < VN; ++j) {
                            const float a = (mmgl_hash32(seed, e0 + j) < thr) ? 0.f : (float)v[j];
                            v[j] = (T)fmaf(keep_scale, a, (float)r[j]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < VN; ++j) v[j] = (T)((float)v[j] + (float)r[j]);
                    }
                    if (sum_out) *(V*)(sum_out + (size_t)row * cols + c * VN) = v;
                }
#pragma unroll
                for (int j = 0; j < VN; ++j) { xv[i][j] = (float)v[j]; s += xv[i][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < VN; ++j) xv[i][j] = 0.f;
            }
        }
        float mu = 0.f;
        if constexpr (!RMS) mu = row_sum<TPR>(s, red, rib) / cols;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tr + i * TPR;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < VN; ++j) { const float d = xv[i][j] - mu; ss += d * d; }
            }
        }
        const float var = row_sum<TPR>(ss, red, rib) / cols;
        const float rs = rsqrtf(var + eps);
        if (live && tr == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tr + i * TPR;
            if (live && c < nchunks) {
                V o;
                V gv, bv;
                if (gamma) gv = *(const V*)(gamma + c * VN);
                if (beta) bv = *(const V*)(beta + c * VN);
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    float t = (xv[i][j] - mu) * rs;
                    if (gamma) t *= (float)gv[j];
                    if (beta) t += (float)bv[j];
                    o[j] = (T)t;
                }
                *(V*)(y + (size_t)row * cols + c * VN) = o;
            }
        }
    }
}

// MODE bit 0: RMSNorm (no mean); bit 1: accumulate the dgamma / dbeta column partials (skipped for frozen affines)
template <typename T, int TPR, int NV, int MODE>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const T* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const T* __restrict__ dres,
                                                       T* __restrict__ dx, T* __restrict__ dx_drop, float* __restrict__ part,
                                                       int rows, int cols, float p, uint64_t seed) {
    constexpr bool RMS = (MODE & 1) != 0, PARAMS = (MODE & 2) != 0;
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    __shared__ float red[4];
    const int rpb = 256 / TPR;
    const int tr = threadIdx.x % TPR, rib = threadIdx.x / TPR;
    const int nchunks = cols / VN;
    float dg[NV][VN], dbt[NV][VN], gm[NV][VN];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = tr + i * TPR;
        V gv;
        const bool hg = gamma && c < nchunks;
        if (hg) gv = *(const V*)(gamma + c * VN);
#pragma unroll
        for (int j = 0; j < VN; ++j) { dg[i][j] = 0.f; dbt[i][j] = 0.f; gm[i][j] = hg ? (float)gv[j] : 1.f; }
    }
    for (int row0 = blockIdx.x * rpb; row0 < rows; row0 += gridDim.x * rpb) {
        const int row = row0 + rib;
        const bool live = row < rows;
        const float mu = (!RMS && live) ? mean[row] : 0.f;
        const float rs = live ? rstd[row] : 0.f;
        float xh[NV][VN], gy[NV][VN];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tr + i * TPR;
            if (live && c < nchunks) {
                const V xv = *(const V*)(x + (size_t)row * cols + c * VN);
                const V dv = *(const V*)(dy + (size_t)row * cols + c * VN);
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    const float h = ((float)xv[j] - mu) * rs;
                    const float d = (float)dv[j];
                    if constexpr (PARAMS) { dg[i][j] += d * h; dbt[i][j] += d; }
                    xh[i][j] = h;
                    gy[i][j] = d * gm[i][j];
                    s1 += gy[i][j];
                    s2 += gy[i][j] * h;
                }
            } else {
#pragma unroll
                for (int j = 0; j < VN; ++j) { xh[i][j] = 0.f; gy[i][j] = 0.f; }
            }
        }
        float m1 = 0.f;
        if constexpr (!RMS) m1 = row_sum<TPR>(s1, red, rib) / cols;
        const float m2 = row_sum<TPR>(s2, red, rib) / cols;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tr + i * TPR;
            if (live && c < nchunks) {
                V o;
                if (dres) {                                   // gradient arriving on the residual stream (x + res) itself
                    const V dr = *(const V*)(dres + (size_t)row * cols + c * VN);
#pragma unroll
                    for (int j = 0; j < VN; ++j) o[j] = (T)(rs * (gy[i][j] - m1 - xh[i][j] * m2) + (float)dr[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < VN; ++j) o[j] = (T)(rs * (gy[i][j] - m1 - xh[i][j] * m2));
                }
                *(V*)(dx + (size_t)row * cols + c * VN) = o;
                if (dx_drop) {                                // gradient of the dropped-out branch x of s = res + dropout(x)
                    const float ks = 1.f / (1.f - p);
                    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
                    const uint64_t e0 = (uint64_t)row * cols + (uint64_t)c * VN;
                    V od;
#pragma unroll
                    for (int j = 0; j < VN; ++j) od[j] = (mmgl_hash32(seed, e0 + j) < thr) ? (T)0.f : (T)((float)o[j] * ks);
                    *(V*)(dx_drop + (size_t)row * cols + c * VN) = od;
                }
            }
        }
    }
    if (PARAMS && part) {
        // rows of one block that share a column are different threads (rib): fold them through LDS-free atomics? no:
        // write one partial row per (block, rib) -> deterministic second pass.
        float* pg = part + ((size_t)(blockIdx.x * rpb + rib)) * cols * 2;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = tr + i * TPR;
            if (c < nchunks) {
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    pg[c * VN + j] = dg[i][j];
                    pg[cols + c * VN + j] = dbt[i][j];
                }
            }
        }
    }
}

// dgamma[c] = sum_p part[p][c], dbeta[c] = sum_p part[p][cols + c].  A partial row is 2*cols floats.
// Block = 32 columns x 8 partial-row lanes (8 independent load streams per column instead of one serial chain),
// folded through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void norm_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int nparts, int cols) {
    __shared__ float red[8][33];
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;               // column in [0, 2*cols)
    float a = 0.f;
    if (c < 2 * cols) {
        int p = py;
        for (; p + 24 < nparts; p += 32) {
            const float v0 = part[(size_t)p * cols * 2 + c], v1 = part[(size_t)(p + 8) * cols * 2 + c];
            const float v2 = part[(size_t)(p + 16) * cols * 2 + c], v3 = part[(size_t)(p + 24) * cols * 2 + c];
            a += (v0 + v1) + (v2 + v3);
        }
        for (; p < nparts; p += 8) a += part[(size_t)p * cols * 2 + c];
    }
    red[py][cx] = a;
    __syncthreads();
    if (py == 0 && c < 2 * cols) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += red[i][cx];
        if (c < cols) { if (dgamma) dgamma[c] = s; }
        else if (dbeta) dbeta[c - cols] = s;
    }
}

struct Geo { int tpr, nv, rpb; };
template <typename T> int geometry(const char* who, int rows, int cols, Geo& g) {
    constexpr int VN = Vec<T>::N;
    MMGL_CHECK_ARG(rows > 0 && cols > 0, "%s: rows/cols must be positive", who);
    if (cols % VN) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: cols=%d must be a multiple of %d", who, cols, VN);
    const int nchunks = cols / VN;
    g.tpr = (nchunks <= 64 * 4) ? 64 : 256;      // one wave per row while the row fits 4 chunks per lane: no block barriers
    g.nv = (nchunks + g.tpr - 1) / g.tpr;
    if (g.nv > NVMAX) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: cols=%d exceeds the register-resident row limit %d", who, cols,
                                256 * NVMAX * VN);
    g.rpb = 256 / g.tpr;
    return MMGL_OK;
}

int bwd_blocks(int rows, int rpb) {
    int b = (rows + rpb - 1) / rpb;
    return b > MAXBLK ? MAXBLK : b;
}

#define NORM_DISPATCH(KERN, T, RMS, g, ...)                                                        \
    do {                                                                                           \
        if (g.tpr == 64) {                                                                         \
            if (g.nv == 1) hipLaunchKernelGGL((KERN<T, 64, 1, RMS>), __VA_ARGS__);                 \
            else if (g.nv == 2) hipLaunchKernelGGL((KERN<T, 64, 2, RMS>), __VA_ARGS__);            \
            else hipLaunchKernelGGL((KERN<T, 64, 4, RMS>), __VA_ARGS__);                           \
        } else {                                                                                   \
            if (g.nv == 1) hipLaunchKernelGGL((KERN<T, 256, 1, RMS>), __VA_ARGS__);                \
            else if (g.nv == 2) hipLaunchKernelGGL((KERN<T, 256, 2, RMS>), __VA_ARGS__);           \
            else hipLaunchKernelGGL((KERN<T, 256, 4, RMS>), __VA_ARGS__);                          \
        }                                                                                          \
    } while (0)

template <typename T, bool RMS>
int norm_fwd(const char* who, const void* x, const void* res, const void* gamma, const void* beta, void* sum_out, void* y,
             float* mean, float* rstd, int rows, int cols, float eps, hipStream_t st, float p = 0.f, uint64_t seed = 0) {
    Geo g;
    int rc = geometry<T>(who, rows, cols, g);
    if (rc) return rc;
    int blocks = (rows + g.rpb - 1) / g.rpb;
    if (blocks > 8192) blocks = 8192;
    NORM_DISPATCH(norm_fwd_kernel, T, RMS, g, dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)res, (const T*)gamma,
                  (const T*)beta, (T*)sum_out, (T*)y, mean, rstd, rows, cols, eps, p, seed);
    MMGL_CHECK_LAUNCH(who);
    return MMGL_OK;
}

template <typename T, bool RMS>
int norm_bwd(const char* who, const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
             const void* dres, void* dx, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int rows, int cols, hipStream_t st,
             void* dx_drop = nullptr, float p = 0.f, uint64_t seed = 0) {
    Geo g;
    int rc = geometry<T>(who, rows, cols, g);
    if (rc) return rc;
    const bool want = dgamma || dbeta;
    int blocks = bwd_blocks(rows, g.rpb);              // capped: every block owns a partial row of column sums
    if (!want) { blocks = (rows + g.rpb - 1) / g.rpb; if (blocks > 8192) blocks = 8192; }
    float* part = nullptr;
    if (want) {
        MMGL_CHECK_ARG(ws && ws_bytes >= (size_t)blocks * g.rpb * cols * 2 * sizeof(float),
                       "%s: workspace too small (%zu B)", who, ws_bytes);
        part = (float*)ws;
    }
    if (want)
        NORM_DISPATCH(norm_bwd_kernel, T, (RMS ? 3 : 2), g, dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)x,
                      (const T*)gamma, mean, rstd, (const T*)dres, (T*)dx, (T*)dx_drop, part, rows, cols, p, seed);
    else
        NORM_DISPATCH(norm_bwd_kernel, T, (RMS ? 1 : 0), g, dim3(blocks), dim3(256), 0, st, (const T*)dy, (const T*)x,
                      (const T*)gamma, mean, rstd, (const T*)dres, (T*)dx, (T*)dx_drop, part, rows, cols, p, seed);
    MMGL_CHECK_LAUNCH(who);
    if (want) {
        hipLaunchKernelGGL(norm_param_reduce_kernel, dim3((2 * cols + 31) / 32), dim3(256), 0, st, part, dgamma, dbeta,
                           blocks * g.rpb, cols);
        MMGL_CHECK_LAUNCH(who);
    }
    return MMGL_OK;
}

}  // namespace

extern "C" size_t mmgl_norm_bwd_workspace(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    // partial rows = blocks * rows_per_block with rows_per_block in {1, 4}: bounded by min(ceil4(rows), 4 * MAXBLK)
    size_t parts = (size_t)((rows + 3) / 4) * 4;
    if (parts > (size_t)MAXBLK * 4) parts = (size_t)MAXBLK * 4;
    return parts * cols * 2 * sizeof(float);
}

extern "C" int mmgl_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                  int rows, int cols, float eps, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && y && mean && rstd, "mmgl_layernorm_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) return norm_fwd<bf16, false>("mmgl_layernorm_fwd", x, nullptr, gamma, beta, nullptr, y, mean, rstd, rows, cols, eps, st);
    if (dtype == MMGL_F32) return norm_fwd<float, false>("mmgl_layernorm_fwd", x, nullptr, gamma, beta, nullptr, y, mean, rstd, rows, cols, eps, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_layernorm_fwd: bad dtype %d", dtype);
}

extern "C" int mmgl_add_layernorm_fwd(const void* x, const void* res, const void* gamma, const void* beta, void* sum_out,
                                      void* y, float* mean, float* rstd, int rows, int cols, float eps, float p_drop,
                                      uint64_t seed, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && res && y, "mmgl_add_layernorm_fwd: null pointer");
    MMGL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mmgl_add_layernorm_fwd: dropout p=%g outside [0,1)", p_drop);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16)
        return norm_fwd<bf16, false>("mmgl_add_layernorm_fwd", x, res, gamma, beta, sum_out, y, mean, rstd, rows, cols, eps, st, p_drop, seed);
    if (dtype == MMGL_F32)
        return norm_fwd<float, false>("mmgl_add_layernorm_fwd", x, res, gamma, beta, sum_out, y, mean, rstd, rows, cols, eps, st, p_drop, seed);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_add_layernorm_fwd: bad dtype %d", dtype);
}

extern "C" int mmgl_add_layernorm_bwd(const void* dy, const void* dsum, const void* sum, const void* gamma, const float* mean,
                                      const float* rstd, void* dres, void* dx, float* dgamma, float* dbeta, void* workspace,
                                      size_t workspace_bytes, int rows, int cols, float p_drop, uint64_t seed, int dtype,
                                      void* stream) {
    MMGL_CHECK_ARG(dy && sum && mean && rstd && dres, "mmgl_add_layernorm_bwd: null pointer");
    MMGL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mmgl_add_layernorm_bwd: dropout p=%g outside [0,1)", p_drop);
    MMGL_CHECK_ARG(p_drop == 0.f || dx, "mmgl_add_layernorm_bwd: dx is required when p_drop > 0");
    hipStream_t st = (hipStream_t)stream;
    void* dxd = p_drop > 0.f ? dx : nullptr;
    if (dtype == MMGL_BF16)
        return norm_bwd<bf16, false>("mmgl_add_layernorm_bwd", dy, sum, gamma, mean, rstd, dsum, dres, dgamma, dbeta, workspace, workspace_bytes, rows, cols, st, dxd, p_drop, seed);
    if (dtype == MMGL_F32)
        return norm_bwd<float, false>("mmgl_add_layernorm_bwd", dy, sum, gamma, mean, rstd, dsum, dres, dgamma, dbeta, workspace, workspace_bytes, rows, cols, st, dxd, p_drop, seed);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_add_layernorm_bwd: bad dtype %d", dtype);
}

extern "C" int mmgl_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                  void* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, int rows,
                                  int cols, int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && x && mean && rstd && dx, "mmgl_layernorm_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16)
        return norm_bwd<bf16, false>("mmgl_layernorm_bwd", dy, x, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, workspace, workspace_bytes, rows, cols, st);
    if (dtype == MMGL_F32)
        return norm_bwd<float, false>("mmgl_layernorm_bwd", dy, x, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, workspace, workspace_bytes, rows, cols, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_layernorm_bwd: bad dtype %d", dtype);
}

extern "C" int mmgl_rmsnorm_fwd(const void* x, const void* gamma, void* y, float* rstd, int rows, int cols, float eps,
                                int dtype, void* stream) {
    MMGL_CHECK_ARG(x && y && rstd, "mmgl_rmsnorm_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) return norm_fwd<bf16, true>("mmgl_rmsnorm_fwd", x, nullptr, gamma, nullptr, nullptr, y, nullptr, rstd, rows, cols, eps, st);
    if (dtype == MMGL_F32) return norm_fwd<float, true>("mmgl_rmsnorm_fwd", x, nullptr, gamma, nullptr, nullptr, y, nullptr, rstd, rows, cols, eps, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_rmsnorm_fwd: bad dtype %d", dtype);
}

extern "C" int mmgl_add_rmsnorm_fwd(const void* x, const void* res, const void* gamma, void* sum_out, void* y, float* rstd, int rows,
                                    int cols, float eps, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && res && sum_out && y && rstd, "mmgl_add_rmsnorm_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) return norm_fwd<bf16, true>("mmgl_add_rmsnorm_fwd", x, res, gamma, nullptr, sum_out, y, nullptr, rstd, rows, cols, eps, st);
    if (dtype == MMGL_F32) return norm_fwd<float, true>("mmgl_add_rmsnorm_fwd", x, res, gamma, nullptr, sum_out, y, nullptr, rstd, rows, cols, eps, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_add_rmsnorm_fwd: bad dtype %d", dtype);
}

extern "C" int mmgl_add_rmsnorm_bwd(const void* dy, const void* dsum, const void* sum, const void* gamma, const float* rstd, void* dres,
                                    float* dgamma, void* workspace, size_t workspace_bytes, int rows, int cols, int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && sum && rstd && dres, "mmgl_add_rmsnorm_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16)
        return norm_bwd<bf16, true>("mmgl_add_rmsnorm_bwd", dy, sum, gamma, nullptr, rstd, dsum, dres, dgamma, nullptr, workspace, workspace_bytes, rows, cols, st);
    if (dtype == MMGL_F32)
        return norm_bwd<float, true>("mmgl_add_rmsnorm_bwd", dy, sum, gamma, nullptr, rstd, dsum, dres, dgamma, nullptr, workspace, workspace_bytes, rows, cols, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_add_rmsnorm_bwd: bad dtype %d", dtype);
}

extern "C" int mmgl_rmsnorm_bwd(const void* dy, const void* x, const void* gamma, const float* rstd, void* dx,
                                float* dgamma, void* workspace, size_t workspace_bytes, int rows, int cols, int dtype,
                                void* stream) {
    MMGL_CHECK_ARG(dy && x && rstd && dx, "mmgl_rmsnorm_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16)
        return norm_bwd<bf16, true>("mmgl_rmsnorm_bwd", dy, x, gamma, nullptr, rstd, nullptr, dx, dgamma, nullptr, workspace, workspace_bytes, rows, cols, st);
    if (dtype == MMGL_F32)
        return norm_bwd<float, true>("mmgl_rmsnorm_bwd", dy, x, gamma, nullptr, rstd, nullptr, dx, dgamma, nullptr, workspace, workspace_bytes, rows, cols, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_rmsnorm_bwd: bad dtype %d", dtype);
}
