#!/bin/bash
# SQ issue/stall counters (one pass) and HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes, kernel-trace only) of gemm8p_kernel.
# usage: tools/pmc_gemm8p.sh [M N K]     (writes gpurun_out/pmc_gemm8p.txt)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/pmc_gemm8p.txt
: > $out
bash tools/pmc_sq.sh "gemm8p" /tmp/sq.txt -- python tools/probes/gemm8p_one.py "$@" > /dev/null 2>&1
cat /tmp/sq.txt >> $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o p -- python tools/probes/gemm8p_one.py "$@" > /tmp/pmc_$ctr.log 2>&1
  python - "$ctr" >> $out <<'PY'
import csv, glob, sys, collections
ctr = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{ctr}/**/*counter_collection.csv", recursive=True)
if not f:
    print(ctr, "no counter csv found"); sys.exit(0)
agg = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if row.get("Counter_Name") == ctr:
        agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if any(t in k for t in ("gemm8p", "copy")):
        print(f"{ctr:10s} avg/dispatch {sum(v)/len(v):14.1f} KiB  x{len(v):4d}  {k[:100]}")
PY
done
tail -3 /tmp/pmc_WRITE_SIZE.log >> $out
cat $out
