// Store acknowledge latency on gfx950: each wave writes 16 x (64 lanes x 16 B) like the GEMM epilogue, then s_waitcnt vmcnt(0);
// clocks between first store and the ack, for several cache-policy (aux) values and numbers of active workgroups.
// hipcc --offload-arch=gfx950 -O3 store_latency.hip -o store_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(512) void k(unsigned short* y, long long* out, int ldy, int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = lane & 15, g = lane >> 4, wr = wave >> 2, wc = wave & 3;
    long long tot = 0;
    for (int r = 0; r < rounds; ++r) {
        // tile (blockIdx, r): 256 rows x 256 cols region
        // MODE (AUX >> 8): 0 new rows every round (new pages), 1 same rows, next 512-byte column block (new lines, same pages), 2 same tile
        const int MODE = AUX >> 8;
        const int tr = MODE == 0 ? blockIdx.x * rounds + r : blockIdx.x, tc = MODE == 1 ? r : 0;
        unsigned short* base = y + (size_t)(tr % 160) * 256 * (size_t)ldy + tc * 256;
        __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
        u32x4 v = {(unsigned)r, (unsigned)lane, 3u, 4u};
        __builtin_amdgcn_s_barrier();
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            unsigned off;
            if (AUX & 64) off = (unsigned)((((wr * 128 + q * 8 + (x & 7)) * ldy) + wc * 64 + 8 * g + 32 * (x >> 3)) * 2);     // 8 rows x 128 B per instruction
            else { const int J = q & 7, T0 = q >> 3; off = (unsigned)((((wr * 128 + J * 16 + x) * ldy) + wc * 64 + 8 * g + 32 * T0) * 2); }   // 16 rows x 64 B
            __builtin_amdgcn_raw_buffer_store_b128(v, d, off, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += __builtin_readcyclecounter() - t0;
        // ~ a tile's worth of idle time between bursts
        for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = tot / rounds;
}

int main() {
    const int ldy = 8192;
    unsigned short* y; long long* out;
    hipMalloc(&y, (size_t)40960 * ldy * 2);
    hipMalloc(&out, 256 * 8 * 8);
    auto run = [&](auto kern, const char* name, int grid) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, y, out, ldy, 10);
        hipDeviceSynchronize();
        std::vector<long long> h(grid * 8);
        hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-10s grid %3d: store->ack clocks (100 MHz counter ticks?) min %lld median %lld max %lld\n", name, grid, h[0], h[h.size() / 2], h.back());
    };
    for (int grid : {1, 16}) {
        run(k<0>, "new pages", grid);
        run(k<256>, "new lines", grid);
        run(k<512>, "same tile", grid);
        run(k<64>, "new pages 128B", grid);
    }
    return 0;
}
