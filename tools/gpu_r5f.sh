#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gemm_nt_gpu.py tests/test_llama_gpu.py tests/test_model_gpu.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for i in 1 2; do
for f in 0 1; do
  MMGL_GEMM_SPLITK_FOLD=$f timeout 300 python bench.py --batch 4 --ref-batch 0 --no-batch-sweep --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold=$f', d['value'], d['ms_per_step'])" >> $O/fold_ab.log
done; done
cat $O/fold_ab.log
for f in 0 1; do
  MMGL_GEMM_SPLITK_FOLD=$f timeout 600 python bench.py --config llama-2-7b --no-cpu-baseline --no-batch-sweep --ref-batch 0 > $O/bench_llama_fold$f.json 2>/dev/null
  python -c "import json; d=json.loads(open('$O/bench_llama_fold$f.json').read().strip().splitlines()[-1]); print('llama fold=$f', d['value'], d['ms_per_step'])"
done
bash tools/profile_bench.sh r5 > $O/profile.log 2>&1
cp gpurun_out/prof_r564/kernel_stats.csv $O/r5_a_kernel_stats.csv; cp gpurun_out/prof_r54/kernel_stats.csv $O/r5_b_kernel_stats.csv
