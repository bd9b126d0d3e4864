// Internal interface of the persistent 256x256 ping-pong GEMM (gemm8p.hip), used by gemm.hip's dispatchers.
#pragma once
#include "common.h"

// act codes: 0 none, 1 relu, 2 gelu (erf), 3 quick_gelu, 4 gelu (tanh)  (= mmgl_activation_fwd's)
// row strides ldx / ldw / ldy in elements (resid and zmask share ldy)
bool gemm8p_supported(int M, int N, int K, int ldx, int ldw, int ldy);
int gemm8p_splits(int M, int N, int K);      // K splits for few-tile outputs (0 = none)
int gemm8p_num_cu();                          // workgroups of a full launch (= CUs of the current device)
size_t gemm8p_split_bytes(int M, int N, int K);   // fp32 partial tiles of the K-split path (caller's scratch)
int launch_gemm8p(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, const bf16* resid,
                  const bf16* zmask, int M, int N, int K, int act, float scale, hipStream_t st, float* part = nullptr,
                  size_t part_bytes = 0, unsigned* bits_out = nullptr, const unsigned* bits_in = nullptr);
// ReLU mask as bits: 8 KiB per 256x256 output tile ([tile][8 waves][64 lanes][4 dwords], lane-private: the kernel that applies
// them has the same tile -> lane mapping as the one that wrote them); written by the act = 1 epilogue, applied by the plain one
inline size_t gemm8p_bits_bytes(int M, int N) { return (size_t)cdiv(M, 256) * cdiv(N, 256) * 8192; }

// weight gradient on the same structure (gemm8p_tt.hip): Out[RB][RA] = scale * sum_m B[m][rb] A[m][ra], both operands k-major
int gemm8p_tt_splits(int RA, int RB, int M);
int launch_gemm8p_tt(const bf16* A, int lda, const bf16* B, int ldb, bf16* Out, float* part, int RA, int RB, int M, int nsplit, float scale,
                     int accumulate, hipStream_t st);

// 128x128 tiles, two workgroups per CU (gemm_mid.hip): few-tile outputs.  Same epilogue contract as launch_gemm8p.
bool gemm_mid_supported(int M, int N, int K, int ldx, int ldw, int ldy);
int launch_gemm_mid(const bf16* X, int ldx, const bf16* W, int ldw, bf16* Y, int ldy, const bf16* bias, const bf16* resid, const bf16* zmask,
                    int M, int N, int K, int act, float scale, hipStream_t st);
