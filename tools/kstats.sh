#!/bin/bash
# usage: tools/kstats.sh <tag> <filter-regex> -- <command...>   : rocprofv3 kernel stats of a command, filtered
tag=$1; filt=$2; shift 3
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag; rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- "$@" > /tmp/prof_$tag.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/prof_$tag/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
import re
for n,c,t,a,p in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if re.search(r"$filt", n):
        print(f"{a:9.2f} us avg  x{c:5d}  {n[:100]}")
PY
