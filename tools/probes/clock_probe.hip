// What does the shader clock do under a sustained matrix-pipe load?  Every workgroup stamps clock64() (s_memtime: shader-clock cycles)
// and wall_clock64() (s_memrealtime: 100 MHz, constant) around a loop of MFMAs (mode 1: 16x16x32 bf16, 4 independent accumulators per wave,
// 8 waves per CU = two per SIMD; mode 0: the same loop with v_fma instead); shader clock = d(clock64) / d(wall_clock64) * 100 MHz.
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o variants/libclockprobe.so tools/probes/clock_probe.hip
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void clock_kernel(int iters, int mode, long long* out, float* sink) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f;
    const long long t0 = clock64(), w0 = wall_clock64();
    if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            f0 = __builtin_fmaf(f0, 1.0001f, 0.5f); f1 = __builtin_fmaf(f1, 1.0001f, 0.5f);
            f2 = __builtin_fmaf(f2, 1.0001f, 0.5f); f3 = __builtin_fmaf(f3, 1.0001f, 0.5f);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (iters < 0) sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
}

extern "C" int clock_probe(int wgs, int iters, int mode, long long* out, void* stream) {
    hipLaunchKernelGGL(clock_kernel, dim3(wgs), dim3(512), 0, (hipStream_t)stream, iters, mode, out, (float*)nullptr);
    return (int)hipGetLastError();
}
