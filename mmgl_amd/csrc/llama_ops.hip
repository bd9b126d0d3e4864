// Elementwise pieces of the frozen Llama-family decoder layers (BASELINE.json config 5: Llama-2-7B), HBM-bound:
//   rotary position embedding applied IN PLACE to the q and k column blocks of a fused-QKV GEMM output, and the SwiGLU product
//   silu(gate) * up over a fused [gate | up] GEMM output, with their backwards.
// The reference's fork is OPT-only (model/modelling_cross_attention.py:278-375 has learned positions and a ReLU FFN); these
// follow the transformers LlamaDecoderLayer the Llama variant loads through the same HF API (rotate_half convention:
// q' = q*cos + rotate_half(q)*sin with rotate_half(x) = [-x[D/2:], x[:D/2]]).
#include "common.h"

namespace {

inline int llama_blocks(size_t nvec) {
    size_t b = (nvec + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}

// buf [rows, ld] ; for each of `nblk` column blocks of H heads x D (q, then k) rotate pairs (i, i + D/2) of every head by the
// angle of position t = row % T:  cs [T, D/2] float2 (cos, sin).  sign = +1 forward, -1 backward (the transpose rotation).
// One thread = 8 consecutive i of one (row, block, head): two 16-byte loads, two 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(const T* src, T* dst, const f32x2* __restrict__ cs, size_t rows, int Tlen, int H, int D,
                                                   int ld, int nblk, int nall, float sign) {
    // src == dst: in place.  Blocks nblk .. nall-1 of a row (v of a fused q | k | v buffer) are copied unrotated (out-of-place calls).
    constexpr int VN = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
    const int half = D / 2, per_head = half / VN;
    const size_t per_row = (size_t)nall * H * per_head, total = rows * per_row;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const size_t row = id / per_row;
        const int rem = (int)(id - row * per_row);
        const int bh = rem / per_head, c = rem - bh * per_head;          // bh = block * H + head
        const int t = (int)(row % (size_t)Tlen);
        const size_t off = row * (size_t)ld + (size_t)bh * D + c * VN;
        V lo = *(const V*)(src + off), hi = *(const V*)(src + off + half);
        if (bh < nblk * H) {
            const f32x2* a = cs + (size_t)t * half + c * VN;
            V olo, ohi;
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const float co = a[e][0], si = a[e][1] * sign;
                const float x0 = (float)lo[e], x1 = (float)hi[e];
                olo[e] = (T)(x0 * co - x1 * si);
                ohi[e] = (T)(x1 * co + x0 * si);
            }
            lo = olo;
            hi = ohi;
        }
        *(V*)(dst + off) = lo;
        *(V*)(dst + off + half) = hi;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// y[M, F] = silu(gu[:, :F]) * gu[:, F:]
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ gu, T* __restrict__ y, size_t M, int F) {
    constexpr int VN = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
    const int fv = F / VN;
    const size_t total = M * (size_t)fv;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const size_t row = id / fv;
        const int c = (int)(id - row * fv);
        const T* p = gu + row * (size_t)(2 * F) + c * VN;
        const V g = *(const V*)p, u = *(const V*)(p + F);
        V o;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float gg = (float)g[e];
            o[e] = (T)(gg * sigmoidf_(gg) * (float)u[e]);
        }
        *(V*)(y + row * (size_t)F + c * VN) = o;
    }
}

// dgu[:, :F] = dy * u * silu'(g),  dgu[:, F:] = dy * silu(g)
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ gu, T* __restrict__ dgu, size_t M,
                                                         int F) {
    constexpr int VN = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(16 / sizeof(T))));
    const int fv = F / VN;
    const size_t total = M * (size_t)fv;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const size_t row = id / fv;
        const int c = (int)(id - row * fv);
        const size_t o2 = row * (size_t)(2 * F) + c * VN;
        const V g = *(const V*)(gu + o2), u = *(const V*)(gu + o2 + F), d = *(const V*)(dy + row * (size_t)F + c * VN);
        V dg, du;
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const float gg = (float)g[e], s = sigmoidf_(gg), dd = (float)d[e];
            dg[e] = (T)(dd * (float)u[e] * s * (1.f + gg * (1.f - s)));
            du[e] = (T)(dd * gg * s);
        }
        *(V*)(dgu + o2) = dg;
        *(V*)(dgu + o2 + F) = du;
    }
}

}  // namespace

static int rope_launch(const char* who, const void* src, void* dst, const float* cos_sin, size_t rows, int T, int H, int D, int ld, int nblk,
                       int nall, int backward, int dtype, void* stream) {
    MMGL_CHECK_ARG(src && dst && cos_sin, "%s: null pointer", who);
    MMGL_CHECK_ARG(T > 0 && H > 0 && nblk > 0 && nall >= nblk && ld >= nall * H * D, "%s: bad sizes (T=%d H=%d D=%d ld=%d blocks %d of %d)", who, T, H, D, ld, nblk, nall);
    const int vn = dtype == MMGL_BF16 ? 8 : 4;
    if (D % (2 * vn) || ld % vn) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "%s: head_dim %d / row stride %d must be multiples of %d", who, D, ld, 2 * vn);
    if (rows == 0) return MMGL_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = rows * (size_t)nall * H * (D / 2 / vn);
    const float sign = backward ? -1.f : 1.f;
    if (dtype == MMGL_BF16)
        hipLaunchKernelGGL(rope_kernel<bf16>, dim3(llama_blocks(total)), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, (const f32x2*)cos_sin, rows, T, H, D, ld, nblk, nall, sign);
    else if (dtype == MMGL_F32)
        hipLaunchKernelGGL(rope_kernel<float>, dim3(llama_blocks(total)), dim3(256), 0, st, (const float*)src, (float*)dst, (const f32x2*)cos_sin, rows, T, H, D, ld, nblk, nall, sign);
    else MMGL_FAIL(MMGL_ERR_INVALID, "%s: bad dtype %d", who, dtype);
    MMGL_CHECK_LAUNCH(who);
    return MMGL_OK;
}

extern "C" int mmgl_rope_inplace(void* buf, const float* cos_sin, size_t rows, int T, int H, int D, int ld, int nblk, int backward,
                                 int dtype, void* stream) {
    return rope_launch("mmgl_rope_inplace", buf, buf, cos_sin, rows, T, H, D, ld, nblk, nblk, backward, dtype, stream);
}

extern "C" int mmgl_rope(const void* src, void* dst, const float* cos_sin, size_t rows, int T, int H, int D, int ld, int nblk, int nall,
                         int backward, int dtype, void* stream) {
    MMGL_CHECK_ARG(src != dst, "mmgl_rope: src == dst (use mmgl_rope_inplace)");
    return rope_launch("mmgl_rope", src, dst, cos_sin, rows, T, H, D, ld, nblk, nall, backward, dtype, stream);
}

extern "C" int mmgl_swiglu_fwd(const void* gate_up, void* y, size_t M, int F, int dtype, void* stream) {
    MMGL_CHECK_ARG(gate_up && y && F > 0, "mmgl_swiglu_fwd: bad arguments");
    const int vn = dtype == MMGL_BF16 ? 8 : 4;
    if (F % vn) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "mmgl_swiglu_fwd: intermediate size %d must be a multiple of %d", F, vn);
    if (M == 0) return MMGL_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = M * (size_t)(F / vn);
    if (dtype == MMGL_BF16) hipLaunchKernelGGL(swiglu_fwd_kernel<bf16>, dim3(llama_blocks(total)), dim3(256), 0, st, (const bf16*)gate_up, (bf16*)y, M, F);
    else if (dtype == MMGL_F32) hipLaunchKernelGGL(swiglu_fwd_kernel<float>, dim3(llama_blocks(total)), dim3(256), 0, st, (const float*)gate_up, (float*)y, M, F);
    else MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_swiglu_fwd: bad dtype %d", dtype);
    MMGL_CHECK_LAUNCH("mmgl_swiglu_fwd");
    return MMGL_OK;
}

extern "C" int mmgl_swiglu_bwd(const void* dy, const void* gate_up, void* dgate_up, size_t M, int F, int dtype, void* stream) {
    MMGL_CHECK_ARG(dy && gate_up && dgate_up && F > 0, "mmgl_swiglu_bwd: bad arguments");
    const int vn = dtype == MMGL_BF16 ? 8 : 4;
    if (F % vn) MMGL_FAIL(MMGL_ERR_UNSUPPORTED, "mmgl_swiglu_bwd: intermediate size %d must be a multiple of %d", F, vn);
    if (M == 0) return MMGL_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = M * (size_t)(F / vn);
    if (dtype == MMGL_BF16)
        hipLaunchKernelGGL(swiglu_bwd_kernel<bf16>, dim3(llama_blocks(total)), dim3(256), 0, st, (const bf16*)dy, (const bf16*)gate_up, (bf16*)dgate_up, M, F);
    else if (dtype == MMGL_F32)
        hipLaunchKernelGGL(swiglu_bwd_kernel<float>, dim3(llama_blocks(total)), dim3(256), 0, st, (const float*)dy, (const float*)gate_up, (float*)dgate_up, M, F);
    else MMGL_FAIL(MMGL_ERR_INVALID, "mmgl_swiglu_bwd: bad dtype %d", dtype);
    MMGL_CHECK_LAUNCH("mmgl_swiglu_bwd");
    return MMGL_OK;
}
