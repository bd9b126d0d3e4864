// Internal interface of the persistent 256x256 ping-pong GEMM (gemm8p.hip), used by gemm.hip's dispatchers.
#pragma once
#include "common.h"

// act codes: 0 none, 1 relu, 2 gelu (erf), 3 quick_gelu, 4 gelu (tanh)  (= mmgl_activation_fwd's)
// row strides ldx / ldw / ldy in elements (resid and zmask share ldy)
bool gemm8p_supported(int M, int N, int K, int ldx, int ldw, int ldy);
int launch_gemm8p(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, const bf16* resid,
                  const bf16* zmask, int M, int N, int K, int act, float scale, hipStream_t st);
