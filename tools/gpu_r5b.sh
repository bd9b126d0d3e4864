#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
MMGL_GEMM_4W=1 timeout 300 python tools/probes/gemm4w_check.py 2>/dev/null | tail -12 >> $O/ablate.log
echo "== m0" >> $O/ablate.log
MMGL_GEMM_4W=1 timeout 300 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/ablate.log
for v in m0a4 m0a16 m0a1 m1; do
  echo "== $v" >> $O/ablate.log
  MMGL_GEMM_4W=1 MMGL_LIB_PATH=variants/lib_g4_$v.so timeout 300 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/ablate.log
done
echo "== m0" >> $O/ablate.log
MMGL_GEMM_4W=1 timeout 300 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/ablate.log
echo "== 8p" >> $O/ablate.log
timeout 300 python tools/probes/gemm4w_check.py time 2>/dev/null >> $O/ablate.log
cat $O/ablate.log
