#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
MMGL_GEMM_8H=1 timeout 180 python tools/probes/gemm4w_check.py > $O/check8h.log 2>&1; tail -2 $O/check8h.log
for v in s1 aux0 aux2 s1aux0; do
  MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_$v.so timeout 180 python tools/probes/gemm4w_check.py 2>&1 | tail -1 | sed "s/^/$v: /"
  echo "== $v" >> $O/time.log; MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_$v.so timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
done
echo "== 8h" >> $O/time.log; MMGL_GEMM_8H=1 timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
echo "== 8p" >> $O/time.log; timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/time.log
cat $O/time.log
