#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
bash tools/profile_bench.sh r5 > $O/profile.log 2>&1
cp gpurun_out/prof_r564/kernel_stats.csv $O/r5_a_kernel_stats.csv; cp gpurun_out/prof_r54/kernel_stats.csv $O/r5_b_kernel_stats.csv
bash tools/pmc_xattn.sh 64 > /dev/null 2>&1; cp gpurun_out/pmc_xattn_B64.txt $O/
bash tools/pmc_xattn.sh 8 llama > /dev/null 2>&1; cp gpurun_out/pmc_xattn_B8llama.txt $O/
head -12 $O/r5_a_kernel_stats.csv | cut -c1-160; cat $O/pmc_xattn_B64.txt | head -20
