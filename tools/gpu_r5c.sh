#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
for i in 1 2 3; do
  for c in 511867a HEAD; do
    if [ $c = HEAD ]; then d=.; else d=_bisect/$c; fi
    (cd $d && timeout 300 python bench.py --config opt-125m --batch 4 --ref-batch 0 --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing $( [ $c = HEAD ] && echo --no-batch-sweep ) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'])" ) >> $O/regress.log
  done
done
cat $O/regress.log
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
