#!/usr/bin/env python
"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV for profiles/."""
import csv
import sqlite3
import sys

import glob
import os

db, out = sys.argv[1], sys.argv[2]
if os.path.isdir(db):                       # the -d directory of the rocprofv3 run
    db = sorted(glob.glob(os.path.join(db, "**", "*.db"), recursive=True))[0]
con = sqlite3.connect(db)
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, round(tot / 1e3, 3) if tot > 1e6 else round(tot, 3), round(avg, 3), round(pct, 4)])
print(f"{out}: {len(rows)} kernels")
