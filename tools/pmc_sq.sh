#!/bin/bash
# SQ issue/stall counters of a command's kernels in ONE rocprofv3 pass (8 SQ slots, kernel-trace only).
# usage: tools/pmc_sq.sh <kernel-name-regex> <out-file> -- <command...>
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles.
filt=$1; out=$2; shift 3
export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
ctrs=${PMC_SQ_COUNTERS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS}
rocprofv3 --kernel-trace --pmc $ctrs \
  --output-format csv -d /tmp/pmc_sq -o p -- "$@" > /tmp/pmc_sq.log 2>&1
python - "$filt" > "$out" <<'PY'
import csv, glob, sys, collections, re
filt = sys.argv[1]
f = glob.glob("/tmp/pmc_sq/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter csv found"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    if re.search(filt, row["Kernel_Name"]):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    wc = sum(cs["SQ_WAVE_CYCLES"]) / max(1, len(cs["SQ_WAVE_CYCLES"]))
    print(k[:120])
    for c, v in sorted(cs.items()):
        a = sum(v) / len(v)
        print(f"    {c:28s} {a:16.0f}  ({a / wc * 100 if wc else 0:6.1f} % of WAVE_CYCLES)  x{len(v)}")
PY
cat "$out"
