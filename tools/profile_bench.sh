#!/bin/bash
# rocprofv3 kernel statistics of the default bench (B = 64) and of the reference's batch (B = 4): gpurun_out/prof_<tag>{64,4}/kernel_stats.csv
# usage: tools/profile_bench.sh <tag> [extra bench args]
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}; shift
for b in 64 4; do
  d=gpurun_out/prof_${tag}${b}; mkdir -p $d
  if [ $b = 64 ]; then args="--steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --ref-batch 0 --no-batch-sweep"; else args="--batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --ref-batch 0 --no-batch-sweep"; fi
  rm -rf /tmp/prof_$tag$b
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag$b -o p -- python bench.py $args "$@" > $d/bench.log 2>&1
  f=$(find /tmp/prof_$tag$b -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $d/kernel_stats.csv; fi
  tail -1 $d/bench.log | cut -c1-200
done
