#!/bin/bash
# xattn_fwd_kernel at the bench shape (B = 64, H = 32, T = 640, S = 64, D = 64): workgroups per (batch, head) (MMGL_XATTN_MIN_CHUNKS)
# x library variants (tools/build_variant.py: 16-row tiles per wave = more, smaller workgroups per CU)
run() { for r in 1 2; do env "$@" python tools/bench_xattn.py 64 2>/dev/null | grep "^B=" | cut -c1-100 | sed "s|^|$* : |"; done; }
for lib in "" variants/lib_qt1.so variants/lib_qt1mw6.so; do
  for c in 1 2 3; do
    if [ -z "$lib" ]; then run MMGL_XATTN_MIN_CHUNKS=$c; else run MMGL_LIB_PATH=$PWD/$lib MMGL_XATTN_MIN_CHUNKS=$c; fi
  done
done
