#!/bin/bash
# HBM traffic of the cross-attention kernels from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only.  Calibration pass: a 256 MiB torch copy (16 B/lane).
# usage: tools/pmc_xattn.sh <B>      (writes gpurun_out/pmc_xattn_B<B>.txt)
B=${1:-8}
SHAPE=${2:-}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/pmc_xattn_B$B$SHAPE.txt
: > $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o p -- python tools/pmc_target.py $B $SHAPE > /tmp/pmc_$ctr.log 2>&1
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
    if any(t in k for t in ("xattn", "reduce_partials", "elementwise", "copy")):
        print(f"{ctr:10s} avg/dispatch {sum(v)/len(v):14.1f}  x{len(v):4d}  {k[:110]}")
PY
done
cat $out
