#!/bin/bash
# the four bench configurations, one JSON line each into gpurun_out/r3/bench_<config>.log
mkdir -p gpurun_out/r3
for c in opt-1.3b opt-125m opt-1.3b-lora llama-2-7b; do
  extra="--no-cpu-baseline"
  [ $c = opt-1.3b ] && extra=""
  python bench.py --config $c $extra 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r3/bench_$c.log
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r3/bench_{c}.log").read())
    k = d.get("kernels", {})
    pick = {n: (round(v.get("ms_avg", 0), 4), v.get("frac")) for n, v in k.items() if n in ("mmgl_selfattn_fwd", "mmgl_selfattn_bwd", "mmgl_encattn_fwd", "mmgl_gemm_nt", "mmgl_linear_bwd", "mmgl_linear_fwd", "mmgl_xattn_fwd")}
    print(c, d["value"], d["unit"], d["ms_per_step"], "ms/step", "roofline", d.get("roofline", {}).get("frac"), "ref-batch", (d.get("at_reference_batch") or {}).get("value"), pick)
except Exception as e:
    print(c, "FAILED", e, open(f"gpurun_out/r3/bench_{c}.log").read()[-500:])
PY
done
