#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
MMGL_LIB_PATH=variants/lib_p8_pertile.so timeout 180 python tools/probes/gemm4w_check.py > $O/check.log 2>&1; tail -2 $O/check.log
for i in 1 2; do
  echo "== pertile" >> $O/time.log; MMGL_LIB_PATH=variants/lib_p8_pertile.so timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/time.log
  echo "== 8p" >> $O/time.log; timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/time.log
done
cat $O/time.log
