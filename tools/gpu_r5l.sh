#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_trace.so timeout 120 python tools/probes/gemm8h_trace.py 2048 > $O/trace2048.log 2>&1
MMGL_GEMM_8H=1 MMGL_LIB_PATH=variants/lib_h8_trace.so timeout 120 python tools/probes/gemm8h_trace.py 768 > $O/trace768.log 2>&1
cat $O/trace2048.log $O/trace768.log
