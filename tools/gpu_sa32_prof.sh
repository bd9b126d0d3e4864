#!/bin/bash
# kernel times (rocprofv3 --stats) + SQ counters of the self-attention kernels at B = 64
mkdir -p gpurun_out/sa32
export TMPDIR=/tmp
rm -rf /tmp/sa32_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sa32_stats -o s -- python tools/probes/selfattn_one.py 64 > /dev/null 2>&1
f=$(find /tmp/sa32_stats -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 "$f" | head -8 | tee gpurun_out/sa32/kernel_stats.txt
bash tools/pmc_sq.sh "sa32" gpurun_out/sa32/sq_B64.txt -- python tools/probes/selfattn_one.py 64 > /dev/null 2>&1
cat gpurun_out/sa32/sq_B64.txt
