#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_general_gpu.py tests/test_gemm_nt_gpu.py -x -q > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
tail -c 1500 $O/bench_default.json
for c in opt-125m opt-1.3b-lora; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
MMGL_GEMM_ROWSPLIT_BAND=0 timeout 600 python bench.py --config opt-1.3b-lora --no-cpu-baseline --no-batch-sweep --ref-batch 0 > $O/bench_lora_noband.json 2>/dev/null
timeout 900 python bench.py --cpu-all-threads --no-batch-sweep --ref-batch 0 --steps 3 > $O/bench_allthreads.json 2>/dev/null
