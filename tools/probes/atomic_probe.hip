// Throughput of fp32 / packed-bf16 global atomic adds in the access pattern a fused attention backward would use for dQ:
// workgroup (b*h, kv tile j) adds a 64x64 tile into rows [64 i, 64 i + 64) x 64 channels of a [B, T, H, D] buffer for i >= j.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_probe.hip -o atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE>
__global__ __launch_bounds__(256) void add_f32(float* acc, int T, int H, int nt) {
    const int bh = blockIdx.x / nt, j = blockIdx.x % nt, b = bh / H, h = bh % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = j; i < nt; ++i) {
        float* base = acc + ((size_t)(b * T + i * 64) * H + h) * 64;
        for (int r = wave; r < 64; r += 4) __hip_atomic_fetch_add(base + (size_t)r * H * 64 + lane, 1.0f, __ATOMIC_RELAXED, SCOPE);
    }
}
__global__ __launch_bounds__(256) void add_pk(uint32_t* acc, int T, int H, int nt) {
    const int bh = blockIdx.x / nt, j = blockIdx.x % nt, b = bh / H, h = bh % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 one = {0x3f80, 0x3f80};
    for (int i = j; i < nt; ++i) {
        uint32_t* base = acc + ((size_t)(b * T + i * 64) * H + h) * 32;
        for (int r = wave * 2 + (lane >> 5); r < 64; r += 8)
            __builtin_amdgcn_global_atomic_fadd_v2bf16((s2 __attribute__((address_space(1)))*)(base + (size_t)r * H * 32 + (lane & 31)), one);
    }
}
__global__ __launch_bounds__(256) void store_f32(float* acc, int T, int H, int nt) {
    const int bh = blockIdx.x / nt, j = blockIdx.x % nt, b = bh / H, h = bh % H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = j; i < nt; ++i) {
        float* base = acc + ((size_t)(b * T + i * 64) * H + h) * 64;
        for (int r = wave; r < 64; r += 4) base[(size_t)r * H * 64 + lane] = 1.0f;
    }
}

int main() {
    const int B = 64, H = 32, T = 640, nt = T / 64;
    const size_t n = (size_t)B * T * H * 64;
    float* acc;
    hipMalloc(&acc, n * 4);
    hipMemset(acc, 0, n * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double adds = (double)B * H * (nt * (nt + 1) / 2) * 4096;
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %8.3f ms  %7.2f G elem/s\n", name, ms, adds / ms * 1e-6);
    };
    run("store f32 (no atomic)", [&] { hipLaunchKernelGGL(store_f32, dim3(B * H * nt), dim3(256), 0, 0, acc, T, H, nt); });
    run("atomic f32 agent scope", [&] { hipLaunchKernelGGL(add_f32<__HIP_MEMORY_SCOPE_AGENT>, dim3(B * H * nt), dim3(256), 0, 0, acc, T, H, nt); });
    run("atomic f32 workgroup scope", [&] { hipLaunchKernelGGL(add_f32<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(B * H * nt), dim3(256), 0, 0, acc, T, H, nt); });
    run("atomic pk bf16", [&] { hipLaunchKernelGGL(add_pk, dim3(B * H * nt), dim3(256), 0, 0, (uint32_t*)acc, T, H, nt); });
    // check: counts of the f32 runs (7 launches each of 2 kernels) on element 0 of the last q tile
    hipMemset(acc, 0, n * 4);
    hipLaunchKernelGGL(add_f32<__HIP_MEMORY_SCOPE_AGENT>, dim3(B * H * nt), dim3(256), 0, 0, acc, T, H, nt);
    std::vector<float> hbuf(64);
    hipMemcpy(hbuf.data(), acc + ((size_t)(0 * T + (nt - 1) * 64) * H + 0) * 64, 256, hipMemcpyDeviceToHost);
    printf("last q tile element after one launch: %.1f (expect %d)\n", hbuf[0], nt);
    return 0;
}
