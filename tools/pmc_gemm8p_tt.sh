cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o p -- python tools/probes/wgrad_one.py 40960 8192 2048 > /tmp/pmc_$ctr.log 2>&1
  python - "$ctr" <<'PY'
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
    print(f"{ctr:10s} avg/dispatch {sum(v)/len(v):14.1f} KiB  x{len(v):4d}  {k[:100]}")
PY
done
