#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the self-attention kernels at config 5's shape, by batch
export TMPDIR=/tmp
python tools/probes/sa_llama.py 2 4 8 12 16 32 2>/dev/null | grep "^B="
for B in 4 8 16; do
  rm -rf /tmp/sa_$B; rocprofv3 --kernel-trace --stats -d /tmp/sa_$B -o p -f csv -- python tools/probes/sa_llama.py $B > /dev/null 2>&1
  f=$(find /tmp/sa_$B -name '*kernel_stats.csv' | head -1)
  echo "== B=$B"; grep "sa32\|selfattn\|rowdot" $f | awk -F'","' '{printf("%10.1f us avg x%s  %s\n", $4/1000, $2, substr($1,2,70))}'
done
