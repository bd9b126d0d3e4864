// Streaming ceiling of the cross-attention core's ACCESS PATTERN, without its arithmetic: copy a [B, T, H*D] bf16 tensor in which
// one workgroup owns `heads` adjacent heads of one sample (a contiguous segment of heads * D * 2 bytes per row, rows 4 KiB apart at
// H * D = 2048) and walks the T rows the way xattn_fwd_kernel does: 4 waves, 16-byte lanes, `rows` rows per wave and step, the
// next step's loads requested before the current step's stores.   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o variants/libsegcopy.so tools/probes/seg_copy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NV> __global__ __launch_bounds__(256) void seg_copy_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int T, int row_chunks,
                                                                         int seg_chunks, int wgs_per_row_set) {
    // row_chunks: 16-byte chunks per full row (H*D*2/16); seg_chunks: chunks of this workgroup's segment; one wave step = 64 lanes * NV chunks
    const int b = blockIdx.x / wgs_per_row_set, sgi = blockIdx.x % wgs_per_row_set;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lanes_per_row = seg_chunks;                       // lanes covering one row's segment
    const int rows_per_step = 64 / lanes_per_row;               // rows one wave instruction covers
    const int r_in = lane / lanes_per_row, c_in = lane % lanes_per_row;
    const size_t base = (size_t)b * T * row_chunks + (size_t)sgi * seg_chunks + c_in;
    const int step_rows = rows_per_step * NV;
    u32x4 cur[NV], nxt[NV];
    int t = wave * step_rows;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int r = t + i * rows_per_step + r_in; cur[i] = (r < T) ? in[base + (size_t)r * row_chunks] : u32x4{0, 0, 0, 0}; }
    for (; t < T; t += 4 * step_rows) {
        const int tn = t + 4 * step_rows;
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int r = tn + i * rows_per_step + r_in; nxt[i] = (r < T) ? in[base + (size_t)r * row_chunks] : u32x4{0, 0, 0, 0}; }
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int r = t + i * rows_per_step + r_in; if (r < T) out[base + (size_t)r * row_chunks] = cur[i]; }
#pragma unroll
        for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
    }
}

extern "C" int seg_copy(const void* in, void* out, int B, int T, int H, int D, int heads, int nv, void* stream) {
    const int row_chunks = H * D * 2 / 16, seg_chunks = heads * D * 2 / 16;
    if (seg_chunks > 64 || 64 % seg_chunks || H % heads) return 1;
    const int per = H / heads;
    dim3 grid(B * per), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (nv == 1) hipLaunchKernelGGL(seg_copy_kernel<1>, grid, block, 0, st, (const u32x4*)in, (u32x4*)out, T, row_chunks, seg_chunks, per);
    else if (nv == 2) hipLaunchKernelGGL(seg_copy_kernel<2>, grid, block, 0, st, (const u32x4*)in, (u32x4*)out, T, row_chunks, seg_chunks, per);
    else if (nv == 4) hipLaunchKernelGGL(seg_copy_kernel<4>, grid, block, 0, st, (const u32x4*)in, (u32x4*)out, T, row_chunks, seg_chunks, per);
    else hipLaunchKernelGGL(seg_copy_kernel<8>, grid, block, 0, st, (const u32x4*)in, (u32x4*)out, T, row_chunks, seg_chunks, per);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
