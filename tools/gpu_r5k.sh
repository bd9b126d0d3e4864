#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
for v in a1 a2 a4 a6; do
  echo "== $v" >> $O/time.log; MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_$v.so timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
done
echo "== 8h" >> $O/time.log; MMGL_GEMM_8H=1 timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
echo "== 8p" >> $O/time.log; timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/time.log
cat $O/time.log
