#!/bin/bash
# Memory-side counters of a command's kernels, one rocprofv3 pass per group (kernel-trace only): FETCH_SIZE | WRITE_SIZE |
# TCC_HIT_sum TCC_MISS_sum.  FETCH_SIZE (KiB) reads 1/2 of the bytes of a 16-B/lane stream on gfx950 (MI355X_MICROARCH.md, HBM).
# usage: tools/pmc_mem.sh <kernel-name-regex> <out-file> -- <command...>
filt=$1; out=$2; shift 3
export TMPDIR=/tmp
: > "$out"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pmc_mem_$tag
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_mem_$tag -o p -- "$@" > /tmp/pmc_mem_$tag.log 2>&1
  python - "$filt" "$tag" >> "$out" <<'PY'
import csv, glob, sys, collections, re
filt, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pmc_mem_{tag}/**/*counter_collection.csv", recursive=True)
if not f:
    print(tag, "no counter csv found"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    if re.search(filt, row["Kernel_Name"]):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    for c, v in sorted(cs.items()):
        print(f"{c:14s} avg/dispatch {sum(v)/len(v):16.1f}  x{len(v):4d}  {k[:100]}")
PY
done
cat "$out"
