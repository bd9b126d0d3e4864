#!/usr/bin/env python
"""gpurun_out/pmc_xattn_B<B>.txt (tools/pmc_xattn.sh) -> profiles/r2_pmc_xattn_B<B>_bf16.json (what bench.py's
roofline.traffic reads).  FETCH_SIZE is doubled (gfx950, 16-B/lane streams; the calibration copy in the same run shows the
1/2), WRITE_SIZE is exact; unit KiB.   usage: python tools/pmc_to_json.py <B> [txt] [json]"""
import json
import re
import sys

B = int(sys.argv[1])
txt = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/pmc_xattn_B{B}.txt"
out = sys.argv[3] if len(sys.argv) > 3 else f"profiles/r3_pmc_xattn_B{B}_bf16.json"
names = {"xattn_fwd_kernel": "xattn_fwd_kernel", "xattn_bwd_dq_kernel": "xattn_bwd_dq_kernel",
         "xattn_bwd_dkv_kernel": "xattn_bwd_dkv_kernel", "xattn_bwd_dkv64_kernel": "xattn_bwd_dkv64_kernel", "xattn_bwd_fused_kernel": "xattn_bwd_fused_kernel", "xattn_bwd_fusedw_kernel": "xattn_bwd_fusedw_kernel", "reduce_partials_kernel": "reduce_partials_kernel",
         "copyBuffer": "calibration_copy"}
k = {}
for line in open(txt):
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE)\s+avg/dispatch\s+([0-9.]+)\s+x\s*(\d+)\s+(.*)", line)
    if not m:
        continue
    for pat, nm in names.items():
        if pat in m.group(4):
            k.setdefault(nm, {})[m.group(1) + "_KiB"] = float(m.group(2))
for nm, d in k.items():
    if "FETCH_SIZE_KiB" in d and "WRITE_SIZE_KiB" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE_KiB"] + d["WRITE_SIZE_KiB"]) * 1024.0
if "reduce_partials_kernel" in k:
    k["reduce_partials_kernel"]["launches_per_bwd"] = 2
doc = {"source": f"tools/pmc_xattn.sh {B}  (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; MI355X, round " + (sys.argv[4] if len(sys.argv) > 4 else "3") + ")",
       "config": {"B": B, "H": 32, "T": 640, "S": 64, "D": 64, "dtype": "bf16"},
       "unit_note": "counter unit = KiB; FETCH_SIZE x2 on gfx950 for 16-B/lane streams (MI355X_MICROARCH.md HBM section); "
                    "calibration_copy = 256 MiB torch copies in the same run (FETCH reads ~1/2 of the bytes, WRITE exact)",
       "kernels": k}
json.dump(doc, open(out, "w"), indent=1)
print(out, {n: round(d.get("hbm_bytes_per_launch", 0) / 1e6, 1) for n, d in k.items()})
