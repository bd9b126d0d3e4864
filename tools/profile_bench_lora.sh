cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_l
rm -rf /tmp/prof_l
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o p -- python bench.py --config opt-1.3b-lora --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --ref-batch 0 > gpurun_out/prof_l/bench.log 2>&1
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/prof_l/kernel_stats.csv; fi
tail -1 gpurun_out/prof_l/bench.log | cut -c1-200
