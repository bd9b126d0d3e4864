#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gemm_nt_gpu.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
