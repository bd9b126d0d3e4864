cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_h gpurun_out/prof_i
for tag in h i; do
  if [ $tag = h ]; then args="--steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --ref-batch 0"; else args="--batch 4 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --ref-batch 0"; fi
  rm -rf /tmp/prof_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py $args > gpurun_out/prof_$tag/bench.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/prof_$tag/kernel_stats.csv; fi
  tail -1 gpurun_out/prof_$tag/bench.log | cut -c1-200
done
