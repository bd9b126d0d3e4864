// The tile boundary of the persistent GEMM in isolation: every wave of a CU issues the 16 output stores of a tile (1 KiB each, 16 rows
// x 64 B at the output's row pitch) and then keeps its LDS-DMA unit stream going (2 loads of 1 KiB per phase).  How long until the
// stores are issued, until the loads queued behind them have landed, and how does that depend on the stores' cache policy?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/store_then_load.hip -o build_probe/store_then_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(512) void k(const char* x, char* y, long long* out, int nstore, int nload, int pitch) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)blockIdx.x * (256u << 10)), 0, 256 << 10, 0x00020000);
    // output tile (block b: column block b % 8, row block b / 8) of a matrix with row pitch `pitch` bytes
    __amdgpu_buffer_rsrc_t dy = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (size_t)(blockIdx.x / 8) * 256 * pitch + (blockIdx.x % 8) * 512), 0, 0x7fffffff, 0x00020000);
    const u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
    // warm the window into L2
    for (int q = 0; q < 32; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (lds_void*)(smem + (q & 7) * 16384 + wave * 1024), 16, (unsigned)(((q * 8 + wave) * 64 + lane) * 16), 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int q = 0; q < nstore; ++q) {
        const unsigned off = (unsigned)(((wave * 32 + (q >> 1) * 16 / 8 * 0 + q * 16 / 16 * 0) ) );      // (kept simple below)
        (void)off;
        const unsigned o = (unsigned)((wave * 32 + (lane >> 2) + 16 * (q & 1)) * pitch + (q >> 1) * 64 + (lane & 3) * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, dy, o, 0, AUX);
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int q = 0; q < nload; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (lds_void*)(smem + (q & 7) * 16384 + wave * 1024), 16, (unsigned)(((q * 8 + wave) * 64 + lane) * 16) & ((256u << 10) - 1), 0, 0, 0);
    const long long t2 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = __builtin_readcyclecounter();
    if (lane == 0) {
        long long* o = out + (blockIdx.x * 8 + wave) * 4;
        o[0] = t1 - t0; o[1] = t2 - t0; o[2] = t3 - t0;
    }
}

int main() {
    char *x, *y; long long* out;
    hipMalloc(&x, (size_t)256 * (256u << 10)); hipMemset(x, 1, (size_t)256 * (256u << 10));
    const int pitch = 4096;
    hipMalloc(&y, (size_t)32 * 256 * pitch + (1 << 20));
    hipMalloc(&out, 256 * 8 * 4 * 8);
    auto run = [&](auto kern, const char* name, int grid, int nstore, int nload) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 131072, 0, x, y, out, nstore, nload, pitch);
        hipDeviceSynchronize();
        std::vector<long long> h(grid * 8 * 4);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<long long> a, b, c;
        for (int i = 0; i < grid * 8; ++i) { a.push_back(h[i * 4]); b.push_back(h[i * 4 + 1]); c.push_back(h[i * 4 + 2]); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
        printf("%-22s grid %3d: %2d stores + %2d loads per wave: stores issued %6lld, loads issued %6lld, all landed %6lld clocks (medians; max %6lld)\n", name, grid, nstore,
               nload, a[a.size() / 2], b[b.size() / 2], c[c.size() / 2], c.back());
    };
    for (int grid : {1, 256}) {
        run(k<0>, "no stores", grid, 0, 12);
        run(k<0>, "aux 0 (write-back)", grid, 16, 12);
        run(k<2>, "aux 2 (nt)", grid, 16, 12);
        run(k<16>, "aux 16 (sc1)", grid, 16, 12);
        run(k<18>, "aux 18 (nt sc1)", grid, 16, 12);
        run(k<3>, "aux 3 (sc0 nt)", grid, 16, 12);
        run(k<18>, "aux 18, stores only", grid, 16, 0);
        run(k<0>, "aux 0, stores only", grid, 16, 0);
    }
    return 0;
}
