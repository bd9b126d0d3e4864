// Stand-in for a co-resident collective kernel: `wgs` workgroups of 256 threads that spin for `clocks` ticks of the 100 MHz wall clock.
// Built as a shared library, driven by tools/probes/gemm_coresident.py on a side stream while GEMMs run on the main one:
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC -o build_probe/liboccupier.so tools/probes/occupier.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void spin_kernel(long long clocks, int* sink) {
    const long long t0 = wall_clock64();
    int k = 0;
    while (wall_clock64() - t0 < clocks) {
        __builtin_amdgcn_s_sleep(32);
        ++k;
    }
    if (clocks < 0) sink[0] = k;
}

extern "C" int occupier_spin(int wgs, long long clocks, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, clocks, (int*)nullptr);
    return (int)hipGetLastError();
}
