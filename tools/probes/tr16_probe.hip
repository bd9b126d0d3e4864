#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void probe(const int* addr_elems, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + addr_elems[l]));
  for (int j = 0; j < 4; ++j) out[l*4+j] = v[j];
}
int main() {
  int h_addr[64]; short h_out[256];
  int *d_addr; short* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  // experiment 1: canonical: lane i in group g reads base g*64 + i*4 (contiguous 4x16 block, row stride 16)
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h_addr[l] = g * 64 + i * 4; }
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_addr, d_out); hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  printf("exp1 canonical (lane: 4 values)\n");
  for (int l = 0; l < 64; ++l) printf("l%02d: %4d %4d %4d %4d%s", l, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3], (l%4==3)?"\n":"   ");
  // experiment 2: row stride 100 elements: lane i -> row (i/4), col block (i%4)*4 ; group g -> rows g*4..
  for (int l = 0; l < 64; ++l) { int g = l >> 4, i = l & 15; h_addr[l] = (g * 4 + i / 4) * 100 + (i % 4) * 4; }
  hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_addr, d_out); hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  printf("exp2 stride100\n");
  for (int l = 0; l < 64; ++l) printf("l%02d: %4d %4d %4d %4d%s", l, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3], (l%4==3)?"\n":"   ");
  return 0;
}
