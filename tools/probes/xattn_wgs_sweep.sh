#!/bin/bash
# xattn_fwd_kernel at the bench shape (B = 64, H = 32, T = 640, S = 64, D = 64) as a function of how many workgroups a (batch, head)
# is cut into (MMGL_XATTN_TARGET_WGS / (B H) = chunks): do shorter, desynchronised workgroups hide the per-workgroup prologue?
for t in 512 4096 6144 8192 12288; do
  echo "== MMGL_XATTN_TARGET_WGS=$t"
  for r in 1 2; do MMGL_XATTN_TARGET_WGS=$t python tools/bench_xattn.py 64 2>/dev/null | grep "^B="; done
done
