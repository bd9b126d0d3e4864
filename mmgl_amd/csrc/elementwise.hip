// HBM-bound elementwise / scatter / reduction kernels of the neighbor-fusion path (gfx950):
//   gated residual (+dropout), neighbor interleave, token cross-entropy, learned-position ids, AdamW.
// All stream 16-byte lane accesses, grid-stride, fp32 math.
#include "common.h"

namespace {

template <typename T> struct Vec {
    static constexpr int N = 16 / sizeof(T);
    typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

inline int stream_blocks(size_t nvec) {
    size_t b = (nvec + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b ? b : 1));
}

// ------------------------------------------------------------------------------------------ activation (frozen encoders)
// One pass over a GEMM output: HF's ACT2FN entries the frozen neighbor encoders use -- RoBERTa "gelu" (erf form),
// CLIP "quick_gelu" x*sigmoid(1.702x) (three eager kernels in torch), "gelu_new"/"gelu_pytorch_tanh", "relu".
template <int ACT> __device__ __forceinline__ float act_apply(float a) {
    if constexpr (ACT == 1) return fmaxf(a, 0.f);
    else if constexpr (ACT == 2) return 0.5f * a * (1.f + erff(a * 0.70710678118654752440f));
    else if constexpr (ACT == 3) return a / (1.f + __expf(-1.702f * a));
    else return 0.5f * a * (1.f + tanhf(0.79788456080286535588f * (a + 0.044715f * a * a * a)));
}
template <typename T, int ACT>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const size_t nvec = n / VN;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const V xv = ((const V*)x)[i];
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) o[j] = (T)act_apply<ACT>((float)xv[j]);
        ((V*)y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * VN; i < n; ++i) y[i] = (T)act_apply<ACT>((float)x[i]);
}

// ------------------------------------------------------------------------------------------ y = x * (*scale)
// The upstream scalar of a loss node (1 / grad_accumulation_steps, reference language_modelling/run_generation.py:483) applied to a
// saved gradient in fp32 with the scalar read from DEVICE memory: one pass, no host sync, no bf16 rounding of the scalar.
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y, size_t n) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const float s = *scale;
    const size_t nvec = n / VN;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const V xv = ((const V*)x)[i];
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) o[j] = (T)((float)xv[j] * s);
        ((V*)y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * VN; i < n; ++i) y[i] = (T)((float)x[i] * s);
}

// ------------------------------------------------------------------------------------------ gated residual
// y = res + tanh(g) * keep(i) * x / (1-p)        (reference modelling_cross_attention.py:332-335, 356-359)
template <typename T>
__global__ __launch_bounds__(256) void gated_fwd_kernel(const T* __restrict__ res, const T* __restrict__ x,
                                                        const float* __restrict__ gate, T* __restrict__ y, size_t n,
                                                        float p, uint64_t seed) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const float tg = gate ? tanhf(*gate) : 1.f;
    const float sc = p > 0.f ? tg / (1.f - p) : tg;
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    const size_t nvec = n / VN;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const V r = ((const V*)res)[i], xv = ((const V*)x)[i];
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            float a = (float)xv[j];
            if (p > 0.f && mmgl_hash32(seed, i * VN + j) < thr) a = 0.f;
            o[j] = (T)fmaf(sc, a, (float)r[j]);
        }
        ((V*)y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * VN; i < n; ++i) {
            float a = (float)x[i];
            if (p > 0.f && mmgl_hash32(seed, i) < thr) a = 0.f;
            y[i] = (T)fmaf(sc, a, (float)res[i]);
        }
}

template <typename T>
__global__ __launch_bounds__(256) void gated_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                        const float* __restrict__ gate, T* __restrict__ dx,
                                                        float* __restrict__ part, size_t n, float p, uint64_t seed) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    __shared__ float red[4];
    const float tg = gate ? tanhf(*gate) : 1.f;
    const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const uint32_t thr = (uint32_t)fminf(p * 4294967296.f, 4294967295.f);
    const size_t nvec = n / VN;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const V d = ((const V*)dy)[i], xv = ((const V*)x)[i];
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            float keep = inv;
            if (p > 0.f && mmgl_hash32(seed, i * VN + j) < thr) keep = 0.f;
            const float dd = (float)d[j];
            acc += dd * keep * (float)xv[j];
            o[j] = (T)(tg * keep * dd);
        }
        ((V*)dx)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nvec * VN; i < n; ++i) {
            float keep = inv;
            if (p > 0.f && mmgl_hash32(seed, i) < thr) keep = 0.f;
            acc += (float)dy[i] * keep * (float)x[i];
            dx[i] = (T)(tg * keep * (float)dy[i]);
        }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void gate_grad_kernel(const float* __restrict__ part, int nparts,
                                                        const float* __restrict__ gate, float* __restrict__ dgate) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
    const float s = block_sum(a, red);
    if (threadIdx.x == 0) {
        const float tg = tanhf(*gate);
        *dgate = (1.f - tg * tg) * s;
    }
}

// ------------------------------------------------------------------------------------------ neighbor interleave
// one block per (b, source neighbor j): copy n_tok*d elements to slot loc[b][j]   (reference :1093-1104)
template <typename T>
__global__ __launch_bounds__(256) void interleave_fwd_kernel(const T* __restrict__ text, const T* __restrict__ vis,
                                                             const int64_t* __restrict__ tloc, const int64_t* __restrict__ iloc,
                                                             const int64_t* __restrict__ tpos, const int64_t* __restrict__ ipos,
                                                             T* __restrict__ out, uint8_t* __restrict__ valid, int Nt, int Ni,
                                                             int n_tok, int d) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const int N = Nt + Ni;
    const int b = blockIdx.x / N, j = blockIdx.x % N;
    const bool is_text = j < Nt;
    const int jj = is_text ? j : j - Nt;
    const int64_t loc = is_text ? tloc[(size_t)b * Nt + jj] : iloc[(size_t)b * Ni + jj];
    const int64_t pos = is_text ? tpos[(size_t)b * Nt + jj] : ipos[(size_t)b * Ni + jj];
    if (loc < 0 || loc >= N) return;
    const size_t len = (size_t)n_tok * d;
    const T* src = is_text ? text + ((size_t)b * Nt + jj) * len : vis + ((size_t)b * Ni + jj) * len;
    T* dst = out + ((size_t)b * N + loc) * len;
    for (size_t i = threadIdx.x; i < len / VN; i += 256) ((V*)dst)[i] = ((const V*)src)[i];
    if (threadIdx.x < n_tok) valid[((size_t)b * N + loc) * n_tok + threadIdx.x] = pos > 0 ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void interleave_bwd_kernel(const T* __restrict__ dout, const int64_t* __restrict__ tloc,
                                                             const int64_t* __restrict__ iloc, T* __restrict__ dtext,
                                                             T* __restrict__ dvis, int Nt, int Ni, int n_tok, int d) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const int N = Nt + Ni;
    const int b = blockIdx.x / N, j = blockIdx.x % N;
    const bool is_text = j < Nt;
    const int jj = is_text ? j : j - Nt;
    const int64_t loc = is_text ? tloc[(size_t)b * Nt + jj] : iloc[(size_t)b * Ni + jj];
    const size_t len = (size_t)n_tok * d;
    T* dst = is_text ? dtext + ((size_t)b * Nt + jj) * len : dvis + ((size_t)b * Ni + jj) * len;
    if (loc < 0 || loc >= N) {
        for (size_t i = threadIdx.x; i < len / VN; i += 256) ((V*)dst)[i] = vzero<V>();
        return;
    }
    const T* src = dout + ((size_t)b * N + loc) * len;
    for (size_t i = threadIdx.x; i < len / VN; i += 256) ((V*)dst)[i] = ((const V*)src)[i];
}

// ------------------------------------------------------------------------------------------ cross entropy
// one block per row: single-pass online (max, sum) per thread, merged across the block.
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     float* __restrict__ row_lse, float* __restrict__ row_loss, int V_,
                                                     int64_t ignore) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    __shared__ float red[4];
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * V_;
    float m = -INFINITY, s = 0.f;
    const int nvec = (((size_t)V_ * sizeof(T)) % 16 == 0) ? V_ / VN : 0;      // odd vocab sizes: rows are not 16-B aligned -> scalar path
    auto fold = [&](const V& v) __attribute__((always_inline)) {
        float lm = (float)v[0];
#pragma unroll
        for (int j = 1; j < VN; ++j) lm = fmaxf(lm, (float)v[j]);
        const float nm = fmaxf(m, lm);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VN; ++j) acc += __expf((float)v[j] - nm);
        s = s * __expf(m - nm) + acc;
        m = nm;
    };
    // four independent 16-byte loads in flight per thread (one load, one wait, nine exps per trip ran at 45 % of the HBM rate)
    int i = threadIdx.x;
    for (; i + 768 < nvec; i += 1024) {
        const V v0 = ((const V*)x)[i], v1 = ((const V*)x)[i + 256], v2 = ((const V*)x)[i + 512], v3 = ((const V*)x)[i + 768];
        fold(v0);
        fold(v1);
        fold(v2);
        fold(v3);
    }
    for (; i < nvec; i += 256) fold(((const V*)x)[i]);
    for (int i = nvec * VN + threadIdx.x; i < V_; i += 256) {
        const float a = (float)x[i];
        const float nm = fmaxf(m, a);
        s = s * __expf(m - nm) + __expf(a - nm);
        m = nm;
    }
    const float gm = block_max(m, red);
    const float gs = block_sum(m == -INFINITY ? 0.f : s * __expf(m - gm), red);
    if (threadIdx.x == 0) {
        const float lse = gm + logf(gs);
        row_lse[row] = lse;
        const int64_t lab = labels[row];
        row_loss[row] = (lab == ignore || lab < 0 || lab >= V_) ? 0.f : lse - (float)x[lab];
    }
}

__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels,
                                                        int rows, int V_, int64_t ignore, float* __restrict__ loss_sum,
                                                        float* __restrict__ count) {
    __shared__ float red[4];
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) {
        const int64_t lab = labels[i];
        if (!(lab == ignore || lab < 0 || lab >= V_)) { a += row_loss[i]; c += 1.f; }
    }
    const float sa = block_sum(a, red);
    const float sc = block_sum(c, red);
    if (threadIdx.x == 0) { *loss_sum = sa; *count = sc; }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ row_lse, const float* __restrict__ count,
                                                     const float* __restrict__ dloss, T* __restrict__ dlogits, int V_,
                                                     int64_t ignore) {
    typedef typename Vec<T>::type V;
    constexpr int VN = Vec<T>::N;
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * V_;
    T* dx = dlogits + (size_t)row * V_;
    const int64_t lab = labels[row];
    const bool ign = (lab == ignore || lab < 0 || lab >= V_);
    const float scale = ign ? 0.f : (*dloss) / fmaxf(*count, 1.f);
    const float lse = row_lse[row];
    const int nvec = (((size_t)V_ * sizeof(T)) % 16 == 0) ? V_ / VN : 0;
    auto grad = [&](int i, const V& v) __attribute__((always_inline)) {
        V o;
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            float g = __expf((float)v[j] - lse);
            if (i * VN + j == lab) g -= 1.f;
            o[j] = (T)(g * scale);
        }
        ((V*)dx)[i] = o;
    };
    int i = threadIdx.x;
    for (; i + 768 < nvec; i += 1024) {                      // four independent loads in flight per thread
        const V v0 = ((const V*)x)[i], v1 = ((const V*)x)[i + 256], v2 = ((const V*)x)[i + 512], v3 = ((const V*)x)[i + 768];
        grad(i, v0);
        grad(i + 256, v1);
        grad(i + 512, v2);
        grad(i + 768, v3);
    }
    for (; i < nvec; i += 256) grad(i, ((const V*)x)[i]);
    for (int i = nvec * VN + threadIdx.x; i < V_; i += 256) {
        float g = __expf((float)x[i] - lse);
        if (i == lab) g -= 1.f;
        dx[i] = (T)(g * scale);
    }
}

// ------------------------------------------------------------------------------------------ learned position ids
// one wave per batch row: inclusive scan of the mask   (reference :135-145)
__global__ __launch_bounds__(64) void position_ids_kernel(const int64_t* __restrict__ mask, int64_t* __restrict__ pos, int T_) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int64_t carry = 0;
    for (int t0 = 0; t0 < T_; t0 += 64) {
        const int t = t0 + lane;
        const int64_t m = t < T_ ? mask[(size_t)b * T_ + t] : 0;
        int64_t v = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t u = __shfl_up(v, o);
            if (lane >= o) v += u;
        }
        if (t < T_) pos[(size_t)b * T_ + t] = (carry + v) * m - 1 + 2;
        carry += __shfl(v, 63);
    }
}

// ------------------------------------------------------------------------------------------ AdamW
// torch.optim.AdamW:  p *= 1 - lr*wd ; m,v update ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T* __restrict__ param, float* __restrict__ master,
                                                    const T* __restrict__ grad, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2s, float gscale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float g = (float)grad[i] * gscale;
        float p = master ? master[i] : (float)param[i];
        p *= 1.f - lr * wd;
        const float mi = b1 * m[i] + (1.f - b1) * g;
        const float vi = b2 * v[i] + (1.f - b2) * g * g;
        m[i] = mi;
        v[i] = vi;
        p -= (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
        if (master) master[i] = p;
        param[i] = (T)p;
    }
}

}  // namespace

#define BY_DTYPE(CALL_BF16, CALL_F32, who)                        \
    do {                                                          \
        if (dtype == MMGL_BF16) { CALL_BF16; }                    \
        else if (dtype == MMGL_F32) { CALL_F32; }                 \
        else MMGL_FAIL(MMGL_ERR_INVALID, "%s: bad dtype %d", who, dtype); \
    } while (0)

extern "C" int mmgl_gated_residual_fwd(const void* residual, const void* x, const float* gate, void* y, size_t n,
                                       float p_drop, uint64_t seed, int dtype, void* stream) {
    MMGL_CHECK_ARG(residual && x && y, "mmgl_gated_residual_fwd: null pointer");
    MMGL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mmgl_gated_residual_fwd: dropout p=%g outside [0,1)", p_drop);
    if (n == 0) return MMGL_OK;
    hipStream_t st = (hipStream_t)stream;
    BY_DTYPE(hipLaunchKernelGGL(gated_fwd_kernel<bf16>, dim3(stream_blocks(n / 8)), dim3(256), 0, st, (const bf16*)residual,
                                (const bf16*)x, gate, (bf16*)y, n, p_drop, seed),
             hipLaunchKernelGGL(gated_fwd_kernel<float>, dim3(stream_blocks(n / 4)), dim3(256), 0, st, (const float*)residual,
                                (const float*)x, gate, (float*)y, n, p_drop, seed),
             "mmgl_gated_residual_fwd");
    MMGL_CHECK_LAUNCH("mmgl_gated_residual_fwd");
    return MMGL_OK;
}

extern "C" size_t mmgl_gated_residual_bwd_workspace(size_t n) { (void)n; return 2048 * sizeof(float); }

extern "C" int mmgl_gated_residual_bwd(const void* dy, const void* x, const float* gate, void* dx, float* dgate,
                                       void* workspace, size_t workspace_bytes, size_t n, float p_drop, uint64_t seed,
                                       int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && x && dx && workspace, "mmgl_gated_residual_bwd: null pointer");
    MMGL_CHECK_ARG(workspace_bytes >= 2048 * sizeof(float), "mmgl_gated_residual_bwd: workspace too small");
    MMGL_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "mmgl_gated_residual_bwd: dropout p=%g outside [0,1)", p_drop);
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    int blocks = stream_blocks(n / (dtype == MMGL_BF16 ? 8 : 4));
    BY_DTYPE(hipLaunchKernelGGL(gated_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, gate,
                                (bf16*)dx, part, n, p_drop, seed),
             hipLaunchKernelGGL(gated_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)x,
                                gate, (float*)dx, part, n, p_drop, seed),
             "mmgl_gated_residual_bwd");
    if (gate && dgate) hipLaunchKernelGGL(gate_grad_kernel, dim3(1), dim3(256), 0, st, part, blocks, gate, dgate);
    MMGL_CHECK_LAUNCH("mmgl_gated_residual_bwd");
    return MMGL_OK;
}

extern "C" int mmgl_neighbor_interleave_fwd(const void* text_emb, const void* vis_emb, const int64_t* text_loc,
                                            const int64_t* img_loc, const int64_t* text_pos, const int64_t* img_pos,
                                            void* out_emb, uint8_t* out_valid, int B, int Nt, int Ni, int n_tok, int d,
                                            int dtype, void* stream) {
    MMGL_CHECK_ARG(B > 0 && Nt >= 0 && Ni >= 0 && Nt + Ni > 0 && n_tok > 0 && d > 0, "mmgl_neighbor_interleave_fwd: bad sizes");
    MMGL_CHECK_ARG(out_emb && out_valid && (Nt == 0 || (text_emb && text_loc && text_pos)) &&
                       (Ni == 0 || (vis_emb && img_loc && img_pos)), "mmgl_neighbor_interleave_fwd: null pointer");
    MMGL_CHECK_ARG(n_tok <= 256, "mmgl_neighbor_interleave_fwd: n_tok > 256");
    const size_t esz = dtype == MMGL_BF16 ? 2 : 4;
    if (((size_t)n_tok * d * esz) % 16) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "mmgl_neighbor_interleave_fwd: n_tok*d must cover whole 16-byte chunks");
    hipStream_t st = (hipStream_t)stream;
    const size_t N = (size_t)Nt + Ni;
    hipMemsetAsync(out_emb, 0, (size_t)B * N * n_tok * d * esz, st);
    hipMemsetAsync(out_valid, 0, (size_t)B * N * n_tok, st);
    BY_DTYPE(hipLaunchKernelGGL(interleave_fwd_kernel<bf16>, dim3(B * (int)N), dim3(256), 0, st, (const bf16*)text_emb,
                                (const bf16*)vis_emb, text_loc, img_loc, text_pos, img_pos, (bf16*)out_emb, out_valid, Nt, Ni,
                                n_tok, d),
             hipLaunchKernelGGL(interleave_fwd_kernel<float>, dim3(B * (int)N), dim3(256), 0, st, (const float*)text_emb,
                                (const float*)vis_emb, text_loc, img_loc, text_pos, img_pos, (float*)out_emb, out_valid, Nt,
                                Ni, n_tok, d),
             "mmgl_neighbor_interleave_fwd");
    MMGL_CHECK_LAUNCH("mmgl_neighbor_interleave_fwd");
    return MMGL_OK;
}

extern "C" int mmgl_neighbor_interleave_bwd(const void* d_out_emb, const int64_t* text_loc, const int64_t* img_loc,
                                            void* d_text_emb, void* d_vis_emb, int B, int Nt, int Ni, int n_tok, int d,
                                            int dtype, void* stream) {
    MMGL_CHECK_ARG(B > 0 && Nt >= 0 && Ni >= 0 && Nt + Ni > 0 && n_tok > 0 && d > 0, "mmgl_neighbor_interleave_bwd: bad sizes");
    MMGL_CHECK_ARG(d_out_emb && (Nt == 0 || (d_text_emb && text_loc)) && (Ni == 0 || (d_vis_emb && img_loc)),
                   "mmgl_neighbor_interleave_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int N = Nt + Ni;
    BY_DTYPE(hipLaunchKernelGGL(interleave_bwd_kernel<bf16>, dim3(B * N), dim3(256), 0, st, (const bf16*)d_out_emb, text_loc,
                                img_loc, (bf16*)d_text_emb, (bf16*)d_vis_emb, Nt, Ni, n_tok, d),
             hipLaunchKernelGGL(interleave_bwd_kernel<float>, dim3(B * N), dim3(256), 0, st, (const float*)d_out_emb, text_loc,
                                img_loc, (float*)d_text_emb, (float*)d_vis_emb, Nt, Ni, n_tok, d),
             "mmgl_neighbor_interleave_bwd");
    MMGL_CHECK_LAUNCH("mmgl_neighbor_interleave_bwd");
    return MMGL_OK;
}

extern "C" int mmgl_cross_entropy_fwd(const void* logits, const int64_t* labels, float* row_lse, float* row_loss,
                                      float* loss_sum, float* count, int rows, int V, int64_t ignore_index, int dtype,
                                      void* stream) {
    MMGL_CHECK_ARG(logits && labels && row_lse && row_loss && loss_sum && count, "mmgl_cross_entropy_fwd: null pointer");
    MMGL_CHECK_ARG(rows > 0 && V > 0, "mmgl_cross_entropy_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    BY_DTYPE(hipLaunchKernelGGL(ce_fwd_kernel<bf16>, dim3(rows), dim3(256), 0, st, (const bf16*)logits, labels, row_lse,
                                row_loss, V, ignore_index),
             hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, st, (const float*)logits, labels, row_lse,
                                row_loss, V, ignore_index),
             "mmgl_cross_entropy_fwd");
    hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, row_loss, labels, rows, V, ignore_index, loss_sum, count);
    MMGL_CHECK_LAUNCH("mmgl_cross_entropy_fwd");
    return MMGL_OK;
}

extern "C" int mmgl_cross_entropy_bwd(const void* logits, const int64_t* labels, const float* row_lse, const float* count,
                                      const float* dloss, void* dlogits, int rows, int V, int64_t ignore_index, int dtype,
                                      void* stream) {
    MMGL_CHECK_ARG(logits && labels && row_lse && count && dloss && dlogits, "mmgl_cross_entropy_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    BY_DTYPE(hipLaunchKernelGGL(ce_bwd_kernel<bf16>, dim3(rows), dim3(256), 0, st, (const bf16*)logits, labels, row_lse, count,
                                dloss, (bf16*)dlogits, V, ignore_index),
             hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(rows), dim3(256), 0, st, (const float*)logits, labels, row_lse,
                                count, dloss, (float*)dlogits, V, ignore_index),
             "mmgl_cross_entropy_bwd");
    MMGL_CHECK_LAUNCH("mmgl_cross_entropy_bwd");
    return MMGL_OK;
}

extern "C" int mmgl_position_ids(const int64_t* attention_mask, int64_t* pos, int B, int T, void* stream) {
    MMGL_CHECK_ARG(attention_mask && pos && B > 0 && T > 0, "mmgl_position_ids: bad arguments");
    hipLaunchKernelGGL(position_ids_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, attention_mask, pos, T);
    MMGL_CHECK_LAUNCH("mmgl_position_ids");
    return MMGL_OK;
}

extern "C" int mmgl_adamw_step(void* param, float* master, const void* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               float grad_scale, int dtype, void* stream) {
    MMGL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step >= 1, "mmgl_adamw_step: bad arguments");
    if (n == 0) return MMGL_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipStream_t st = (hipStream_t)stream;
    const int blocks = stream_blocks(n);
    BY_DTYPE(hipLaunchKernelGGL(adamw_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (bf16*)param, master, (const bf16*)grad,
                                exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale),
             hipLaunchKernelGGL(adamw_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)param, master, (const float*)grad,
                                exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale),
             "mmgl_adamw_step");
    MMGL_CHECK_LAUNCH("mmgl_adamw_step");
    return MMGL_OK;
}

template <typename T> static int act_launch(const void* x, void* y, size_t n, int act, hipStream_t st) {
    const int blocks = stream_blocks(n / (16 / sizeof(T)));
    switch (act) {
        case 1: hipLaunchKernelGGL((act_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)y, n); break;
        case 2: hipLaunchKernelGGL((act_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)y, n); break;
        case 3: hipLaunchKernelGGL((act_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)y, n); break;
        default: hipLaunchKernelGGL((act_kernel<T, 4>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)y, n); break;
    }
    MMGL_CHECK_LAUNCH("mmgl_activation_fwd");
    return MMGL_OK;
}

extern "C" int mmgl_activation_fwd(const void* x, void* y, size_t n, int act, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && y, "mmgl_activation_fwd: null pointer");
    MMGL_CHECK_ARG(act >= 1 && act <= 4, "mmgl_activation_fwd: act %d not in {1 relu, 2 gelu, 3 quick_gelu, 4 gelu_tanh}", act);
    if (n == 0) return MMGL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) return act_launch<bf16>(x, y, n, act, st);
    if (dtype == MMGL_F32) return act_launch<float>(x, y, n, act, st);
    MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_activation_fwd: bad dtype %d", dtype);
}

extern "C" int mmgl_scale(const void* x, const float* scale, void* y, size_t n, int dtype, void* stream) {
    MMGL_CHECK_ARG(x && y && scale, "mmgl_scale: null pointer");
    if (n == 0) return MMGL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMGL_BF16) hipLaunchKernelGGL(scale_kernel<bf16>, dim3(stream_blocks(n / 8)), dim3(256), 0, st, (const bf16*)x, scale, (bf16*)y, n);
    else if (dtype == MMGL_F32) hipLaunchKernelGGL(scale_kernel<float>, dim3(stream_blocks(n / 4)), dim3(256), 0, st, (const float*)x, scale, (float*)y, n);
    else MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_scale: bad dtype %d", dtype);
    MMGL_CHECK_LAUNCH("mmgl_scale");
    return MMGL_OK;
}
