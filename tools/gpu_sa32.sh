#!/bin/bash
# one GPU call: parity of the self-attention / encoder kernels, micro-benchmark across library variants
mkdir -p gpurun_out/sa32
if [ "$1" != "notest" ]; then
python -m pytest tests/test_selfattn_gpu.py tests/test_encoders_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/sa32/pytest.log
cat gpurun_out/sa32/pytest.log
fi
BENCH_SA_LLAMA=1 bash tools/gpu_variants.sh python tools/bench_selfattn.py 64 2>&1 | tee gpurun_out/sa32/bench_variants.log
