#!/bin/bash
# xattn_bwd_fusedw_kernel with two vs three register sets of Q / dO rows in flight (tools/build_variant.py xw3 ... -DXW_SETS=3)
for r in 1 2; do
  echo "== base (XW_SETS=2)"; python tools/bench_xattn.py llama 2>/dev/null | grep "^B="
  echo "== XW_SETS=3"; MMGL_LIB_PATH=$PWD/variants/lib_xw3.so python tools/bench_xattn.py llama 2>/dev/null | grep "^B="
done
echo "== parity of the variant"; MMGL_LIB_PATH=$PWD/variants/lib_xw3.so python -m pytest tests/test_xattn_gpu.py tests/test_llama_gpu.py -m gpu -x -q 2>&1 | tail -2
