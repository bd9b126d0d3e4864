// Per-CU throughput of the global -> LDS path (buffer_load ... lds, 16 B per lane) that feeds the persistent GEMM's unit ring, with
// and without stores in the same queue:  is the 30 B/clk the GEMM's steady state asks of it (16 KiB per 536-clock phase) close to
// what the path can deliver?     hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_rate.hip -o build_probe/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// each wave issues n LDS-DMA loads of 1 KiB (lane-linear) from a `span`-byte window (L2 / MALL resident when small), into a 8-slot
// ring of 16 KiB units like the GEMM's; MIX: every 8th instruction slot of a wave is a 1 KiB store burst instead (nt + sc1)
template <int MIX>
__global__ __launch_bounds__(512) void k(const char* x, char* y, long long* out, int n, unsigned span) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)blockIdx.x * span), 0, (int)span, 0x00020000);
    __amdgpu_buffer_rsrc_t dy = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)blockIdx.x * (8u << 20)), 0, 8 << 20, 0x00020000);
    const u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    unsigned off = (unsigned)((wave * 64 + lane) * 16);
    for (int q = 0; q < n; ++q) {
        if (MIX && (q & 7) == 7) {
            __builtin_amdgcn_raw_buffer_store_b128(v, dy, (unsigned)(((q * 8 + wave) * 64 + lane) * 16) & ((8u << 20) - 1), 0, 18);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (lds_void*)(smem + ((q & 7) * 16384) + wave * 1024), 16, off, 0, 0, 0);
        }
        off += 8192;
        if (off >= span) off -= span;
        if ((q & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // ~one ring of requests in flight per wave
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    char *x, *y; long long* out;
    const size_t xs = (size_t)256 * (4u << 20);
    hipMalloc(&x, xs); hipMemset(x, 1, xs);
    hipMalloc(&y, (size_t)256 * (8u << 20));
    hipMalloc(&out, 256 * 8 * 8);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    auto run = [&](auto kern, const char* name, int grid, int n, unsigned span) {
        hipMemset(out, 0, 256 * 8 * 8);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, x, y, out, n, span);
        hipDeviceSynchronize();
        std::vector<long long> h(grid * 8);
        hipMemcpy(h.data(), out, grid * 8 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%-34s grid %3d window %5u KiB: clocks min %7lld median %7lld max %7lld -> %.1f B/clk per CU (median)\n", name, grid, span >> 10, h[0],
               h[h.size() / 2], h.back(), 8.0 * n * 1024 / h[h.size() / 2]);
    };
    for (int grid : {1, 32, 256}) {
        run(k<0>, "LDS-DMA loads only", grid, 512, 256u << 10);           // 256 KiB window per CU: L2 hits after the first pass
        run(k<0>, "LDS-DMA loads only", grid, 512, 4u << 20);             // 4 MiB window per CU: 1 GiB chip-wide -> HBM / MALL
        run(k<1>, "7 loads : 1 store (nt sc1)", grid, 512, 256u << 10);
    }
    return 0;
}
