// Internal interface of selfattn32.hip (bf16, head_dim 64 / 128 flash attention on 32x32x16 MFMAs); called by the C ABI entry
// points in selfattn.hip.  Not part of the exported ABI (hidden visibility).
#pragma once
#include "common.h"

#define SA32_HIDDEN __attribute__((visibility("hidden")))

// true when the 32x32 kernels cover this problem (bf16 is implied by the caller); Tk = number of key rows
SA32_HIDDEN bool sa32_supported(int D, int Tk);

// causal (+ P prefix keys) forward; same argument meaning as mmgl_selfattn_prefix_fwd
SA32_HIDDEN int sa32_fwd(const void* q, const void* k, const void* v, const uint8_t* valid, void* out, float* lse, int B, int H, int T,
                         int P, int D, int ldq, int ldk, hipStream_t st);
// packed bidirectional forward; same argument meaning as mmgl_encattn_fwd
SA32_HIDDEN int sa32_enc_fwd(const void* q, const void* k, const void* v, const int* cu, void* out, int nseq, int H, int D, int ld_in,
                             int ld_out, int max_len, int q_rows, hipStream_t st);
// backward: parts & 1 = the dQ kernel (also writes delta [B,H,T] = rowsum(dO * O)), parts & 2 = the dK / dV kernel (reads delta).
// Argument meaning as mmgl_selfattn_prefix_bwd: ldg / ldgk are the row strides of dq / dk, dv.
SA32_HIDDEN int sa32_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
                         const uint8_t* valid, void* dq, void* dk, void* dv, float* delta, int B, int H, int T, int P, int D, int ldq,
                         int ldk, int ldg, int ldgk, int parts, hipStream_t st);
