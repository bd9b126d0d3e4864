#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
MMGL_GEMM_8H=1 timeout 180 python tools/probes/gemm4w_check.py > $O/check8h.log 2>&1; echo "rc $?" >> $O/check8h.log; tail -3 $O/check8h.log
MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_trace.so timeout 120 python tools/probes/gemm8h_trace.py 2048 2>/dev/null | grep -v "^   c0\|^   idle" > $O/trace2048.log; cat $O/trace2048.log
if grep -q "ALL OK" $O/check8h.log; then
  echo "== 8h" >> $O/time.log; MMGL_GEMM_8H=1 timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
  echo "== 8h prio0" >> $O/time.log; MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_p0.so timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
  echo "== 8p" >> $O/time.log; timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/time.log
  echo "== 8h" >> $O/time.log; MMGL_GEMM_8H=1 timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^8p/8h/' >> $O/time.log
  cat $O/time.log
fi
