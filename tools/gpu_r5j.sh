#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
MMGL_GEMM_8H=1 timeout 180 python tools/probes/gemm4w_check.py > $O/check8h.log 2>&1; echo "rc $?" >> $O/check8h.log; tail -14 $O/check8h.log
if grep -q "ALL OK" $O/check8h.log; then
for i in 1 2; do
  echo "== 8h" >> $O/time.log; MMGL_GEMM_8H=1 timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null | sed 's/^4w/8h/; s/^8p/8h/' >> $O/time.log
  echo "== 8p" >> $O/time.log; timeout 200 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/time.log
done
cat $O/time.log
fi
