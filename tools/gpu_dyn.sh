#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_gemm_nt_gpu.py -x -q -m gpu -k "dynamic" 2>&1 | grep -v amdgpu.ids | tail -8
python tools/probes/gemm_coresident.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/gemm_coresident.txt
python tools/bench_gemm8p.py 2>&1 | grep -v amdgpu.ids | tail -14
