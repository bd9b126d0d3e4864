#!/usr/bin/env python
"""What bounds the short-K GEMMs of config 2 (d = 768)?  Times mmgl_gemm_nt (persistent 256x256 ping-pong kernel, bias epilogue) at
M = 40960 over a grid of (N, K), turns each time into time PER 256x256 TILE PER CU, and fits   t_tile(K) = a K + X   per N:
  a  = per-K-element cost (ideal: 2*256*256 flop / 4069 flop/clk/CU at 2.4 GHz = 32.2 clk = 13.4 ns per K element at the nominal clock)
  X  = the per-tile boundary (epilogue: convert + store 128 KiB, refill of the operand ring) -- independent of K.
The operand stream a tile needs is 1024 K bytes (256 rows of X + 256 rows of W, bf16) -- divided by the measured tile time it is the
L2->LDS rate the LDS-DMA path actually sustains, to hold against the 63.6 B/clk/CU that path was measured to deliver.
    python tools/probes/gemm_tile_model.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mmgl_amd import ops  # noqa: E402

CLK = 2.4e9
CUS = torch.cuda.get_device_properties(0).multi_processor_count
IDEAL_FLOP_PER_CLK_CU = 2.5e15 / 256 / CLK


def time_gemm(M, N, K, iters=20):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g).bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(x, W, b, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            ops.gemm_nt(x, W, b, out=y)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / iters)
    return best


def main():
    M = 40960
    print(f"{CUS} CUs, nominal {CLK / 1e9} GHz, ideal {IDEAL_FLOP_PER_CLK_CU:.0f} flop/clk/CU; M = {M}")
    print(f"{'N':>6} {'K':>6} {'us':>9} {'TF':>8} {'frac':>6} {'tiles':>6} {'rounds':>6} {'us/tile':>8} {'clk/tile':>9} {'ideal clk':>9} {'DMA B/clk/CU':>12} {'of 63.6':>7}")
    rows = {}
    for N in (768, 2048, 3072, 8192):
        for K in (768, 2048, 3072, 8192):
            t = time_gemm(M, N, K)
            tiles = (M // 256) * ((N + 255) // 256)
            rounds = -(-tiles // CUS)
            t_tile = t / (tiles / CUS)            # average time a CU spends per tile (load-balanced persistent schedule)
            clk = t_tile * CLK
            ideal = 2.0 * 256 * 256 * K / IDEAL_FLOP_PER_CLK_CU
            tf = 2.0 * M * N * K / t / 1e12
            dma = 1024.0 * K / clk
            rows.setdefault(N, []).append((K, t_tile))
            print(f"{N:6d} {K:6d} {t * 1e6:9.1f} {tf:8.1f} {tf / 2500:6.3f} {tiles:6d} {rounds:6d} {t_tile * 1e6:8.2f} {clk:9.0f} {ideal:9.0f} {dma:12.1f} {dma / 63.6:7.2f}")
    print("\nfit t_tile(K) = a K + X per N (least squares over the four K):")
    for N, pts in rows.items():
        ks = torch.tensor([p[0] for p in pts], dtype=torch.float64)
        ts = torch.tensor([p[1] for p in pts], dtype=torch.float64)
        A = torch.stack([ks, torch.ones_like(ks)], 1)
        sol = torch.linalg.lstsq(A, ts[:, None]).solution.flatten()
        a, X = float(sol[0]), float(sol[1])
        ideal_a = 2.0 * 256 * 256 / IDEAL_FLOP_PER_CLK_CU / CLK
        line = f"  N={N:5d}: a = {a * 1e9:6.2f} ns per K element (ideal {ideal_a * 1e9:.2f}: MFMA pipe at {ideal_a / a:.2f} of nominal), X = {X * 1e6:6.2f} us = {X * CLK:7.0f} clk per tile;"
        line += "  ceiling a K/(a K + X) x pipe: " + ", ".join(f"K={k}: {ideal_a * k / (a * k + X):.3f}" for k in (768, 2048, 3072, 8192))
        print(line)


if __name__ == "__main__":
    main()
