#!/bin/bash
# One parameterised GPU-box script (replaces the per-experiment tools/gpu_r*.sh of earlier rounds).  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh <tag> <task> [<task> ...]'
# Output lands under gpurun_out/<tag>/.  Tasks:
#   tests[:<pytest -k expr>]   pytest tests -m gpu (optionally filtered)          -> tests.log
#   file:<tests/test_x.py>     one test file                                        -> tests_<name>.log
#   smoke                      __graft_entry__.smoke()                              -> smoke.log
#   bench[:<config>[:<extra bench.py flags, comma separated>]]                      -> bench_<config>.json (the JSON line) + .log
#   prof[:<config>[:<flags>]]  rocprofv3 --kernel-trace --stats of a short bench    -> prof_<config>_kernel_stats.csv
#   py:<script>[:<args,comma separated>]   python <script> args                     -> <script basename>.log
#   sh:<command with + for spaces>         arbitrary shell                          -> sh.log
set -u
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
for task in "$@"; do
  IFS=: read -r kind a b <<< "$task"
  case $kind in
    tests)  timeout 2400 python -m pytest tests -m gpu -x -q ${a:+-k "$a"} > $O/tests.log 2>&1; echo "[tests] rc $? $(tail -1 $O/tests.log)";;
    file)   n=$(basename $a .py); timeout 1800 python -m pytest $a -m gpu -x -q ${b:+-k "$b"} > $O/tests_$n.log 2>&1; echo "[file $n] rc $? $(tail -1 $O/tests_$n.log)";;
    smoke)  timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "[smoke] rc $? $(tail -1 $O/smoke.log)";;
    bench)  c=${a:-opt-1.3b}; timeout 1500 python bench.py --config $c ${b//,/ } > $O/bench_$c.log 2>&1; rc=$?
            grep '^{"metric"' $O/bench_$c.log | tail -1 > $O/bench_$c.json
            python - $O/bench_$c.json $rc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    k = d.get("kernels", {})
    names = ("mmgl_selfattn_fwd", "mmgl_selfattn_bwd", "mmgl_gemm_nt", "mmgl_linear_bwd", "mmgl_linear_fwd", "mmgl_xattn_fwd", "mmgl_xattn_bwd")
    pick = {n: (round(v.get("ms_avg", 0), 4), v.get("frac")) for n, v in k.items() if n in names}
    pr = {n: v.get("value") for n, v in (d.get("at_reference_protocol") or {}).items() if isinstance(v, dict)}
    print("[bench]", d["config"]["workload"].split()[0], d["value"], d["unit"], d["ms_per_step"], "ms/step roofline", d.get("roofline", {}).get("frac"),
          "ref-batch", (d.get("at_reference_batch") or {}).get("value"), "protocol", pr, "xattn layers", d.get("cross_attention_layers"), pick)
except Exception as e:
    print("[bench] FAILED rc", sys.argv[2], e)
PY
            ;;
    prof)   c=${a:-opt-1.3b}; rm -rf /tmp/prof_$c
            timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o p -f csv -- python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --no-batch-sweep --no-protocol --ref-batch 0 ${b//,/ } > $O/prof_$c.log 2>&1
            f=$(find /tmp/prof_$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/prof_${c}_kernel_stats.csv && head -12 $O/prof_${c}_kernel_stats.csv | cut -c1-160;;
    py)     n=$(basename $a .py); timeout 1500 python $a ${b//,/ } > $O/$n.log 2>&1; echo "[py $n] rc $?"; tail -40 $O/$n.log;;
    sh)     timeout 1500 bash -c "${a//+/ }" > $O/sh.log 2>&1; echo "[sh] rc $?"; tail -20 $O/sh.log;;
    *)      echo "unknown task $task";;
  esac
done
