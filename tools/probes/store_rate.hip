// Per-CU store throughput by access width: one workgroup of 8 waves, each wave issues N stores of `W` bytes per lane to
// distinct lines, then waits; clocks from the first store to the last wave's ack.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// PAT: rows per instruction (1 = contiguous 1 KB, 2 = 2 x 512 B, 4 = 4 x 256 B, 8 = 8 x 128 B, 16 = 16 x 64 B), row pitch in bytes
template <int ROWS>
__global__ __launch_bounds__(512) void kp(char* y, long long* out, int n, int pitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // block b writes the 256 x 512 B tile (b % 32 column block, b / 32 row block) of a matrix with the given pitch
    __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)(blockIdx.x / 32) * 256 * pitch + (blockIdx.x % 32) * 512), 0, 0x7fffffff, 0x00020000);
    u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
    constexpr int LPR = 64 / ROWS;                 // lanes per row
    const long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < n; ++q) {
        // instruction (wave, q) covers rows [(wave*n + q)*ROWS, +ROWS) of a column block 0
        const unsigned off = (unsigned)(((wave * n + q) * ROWS + lane / LPR) * pitch + (lane % LPR) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, d, off, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int W, int WAVES>
__global__ __launch_bounds__(512) void k(char* y, long long* out, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= WAVES) return;
    __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)blockIdx.x * (64 << 20)), 0, 0x7fffffff, 0x00020000);
    u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
    const long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < n; ++q) {
        const unsigned off = (unsigned)(((wave * n + q) * 64 + lane) * W);          // fully contiguous per instruction
        if (W == 16) __builtin_amdgcn_raw_buffer_store_b128(v, d, off, 0, 0);
        else if (W == 8) __builtin_amdgcn_raw_buffer_store_b64(u32x2{v[0], v[1]}, d, off, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(v[0], d, off, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    char* y; long long* out;
    hipMalloc(&y, (size_t)256 * (64 << 20) / 16);
    hipMalloc(&out, 256 * 8 * 8);
    auto run = [&](auto kern, const char* name, int waves, int n, int W) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(512), 0, 0, y, out, n);
        hipDeviceSynchronize();
        std::vector<long long> h(8);
        hipMemcpy(h.data(), out, 8 * 8, hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < waves; ++i) mx = std::max(mx, h[i]);
        printf("%-14s waves %d, %3d stores each: %6lld clocks  -> %.1f B/clk per CU\n", name, waves, n, mx, (double)waves * n * 64 * W / mx);
    };
    run(k<16, 8>, "b128", 8, 64, 16);
    run(k<8, 8>, "b64", 8, 64, 8);
    run(k<4, 8>, "b32", 8, 64, 4);
    run(k<16, 1>, "b128 1 wave", 1, 64, 16);
    run(k<16, 4>, "b128 4 waves", 4, 64, 16);
    run(k<16, 8>, "b128 x256", 8, 256, 16);
    auto runp = [&](auto kern, const char* name, int grid, int n, int pitch) {
        hipMemset(out, 0, 256 * 8 * 8);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, y, out, n, pitch);
        hipDeviceSynchronize();
        std::vector<long long> h(grid * 8);
        hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-16s grid %3d pitch %6d: clocks to ack: min %6lld median %6lld max %6lld -> %.1f B/clk per CU at the slowest\n", name, grid, pitch, h[0], h[h.size() / 2], h.back(), 8.0 * n * 1024 / h.back());
    };
    for (int grid : {1, 8, 32, 64, 128, 256}) runp(kp<16>, "16 rows x 64 B", grid, 16, 16384);
    for (int grid : {32, 256}) runp(kp<2>, "2 rows x 512 B", grid, 16, 16384);
    return 0;
}
