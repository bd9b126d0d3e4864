#!/bin/bash
# xattn_bwd_fused_kernel (config 3's shape) with two vs three register sets of Q / dO rows in flight: stand-alone and inside the bench step
for r in 1 2; do
  echo "== base"; python tools/bench_xattn.py 64 2>/dev/null | grep "^B="
  echo "== XF_SETS=3"; MMGL_LIB_PATH=$PWD/variants/lib_xf3.so python tools/bench_xattn.py 64 2>/dev/null | grep "^B="
done
MMGL_LIB_PATH=$PWD/variants/lib_xf3.so python -m pytest tests/test_xattn_gpu.py -m gpu -x -q 2>&1 | tail -1
bash tools/probes/ab_bench.sh xf3
