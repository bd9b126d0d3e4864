#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
bash tools/profile_bench.sh r5 > $O/profile.log 2>&1
cp gpurun_out/prof_r564/kernel_stats.csv $O/r5_a_kernel_stats.csv; cp gpurun_out/prof_r54/kernel_stats.csv $O/r5_b_kernel_stats.csv
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['batch_sweep'], d['roofline'])"
