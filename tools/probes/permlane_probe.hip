#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void swap16(float v, float& lo, float& hi) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    // inline asm, not __builtin_amdgcn_permlane16_swap: this clang maps BOTH result elements of the builtin to
    // extractvalue 0 (tools/probes/permlane_probe.hip), silently returning the lower rows twice.  s_nop 1 = the two wait
    // states a VALU write needs before a v_permlane read.  After the swap: u = {r0, r0', r2, r2'}, w = {r1, r1', r3, r3'}.
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
    lo = __builtin_bit_cast(float, u);
    hi = __builtin_bit_cast(float, w);
}
__device__ __forceinline__ void swap32(float v, float& lo, float& hi) {
    unsigned u = __builtin_bit_cast(unsigned, v), w = u;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
    lo = __builtin_bit_cast(float, u);
    hi = __builtin_bit_cast(float, w);
}
__global__ void probe(float* out) {
  int l = threadIdx.x;
  float v = (float)l, a, b;
  swap16(v, a, b); out[l] = a; out[64 + l] = b;
  float s = a + b;
  swap32(s, a, b); out[128 + l] = a; out[192 + l] = b;
  out[256 + l] = a + b;
}
int main() {
  float* d; float h[320];
  hipMalloc(&d, sizeof(h));
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"p16.lo", "p16.hi", "p32.lo", "p32.hi", "sum   "};
  for (int k = 0; k < 5; ++k) { printf("%s:", names[k]); for (int l = 0; l < 64; l += 4) printf(" %3.0f", h[k * 64 + l]); printf("\n"); }
  return 0;
}
