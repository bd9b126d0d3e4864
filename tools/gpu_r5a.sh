#!/bin/bash
# round 5, first GPU call: the four-wave GEMM's correctness + timing, and the config-2 B = 4 bisect (r3 end .. HEAD)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
MMGL_GEMM_4W=1 timeout 600 python tools/probes/gemm4w_check.py > $O/check4w.log 2>&1
echo "check rc $?" >> $O/check4w.log
MMGL_GEMM_4W=1 timeout 300 python tools/probes/gemm4w_check.py time > $O/time4w.log 2>&1
timeout 300 python tools/probes/gemm4w_check.py time > $O/time8p.log 2>&1
MMGL_GEMM_4W=1 timeout 300 python tools/probes/gemm4w_check.py time >> $O/time4w.log 2>&1
timeout 300 python tools/probes/gemm4w_check.py time >> $O/time8p.log 2>&1
for c in 511867a 4741879 8ba3dad 51cc750; do
  (cd _bisect/$c && timeout 300 python bench.py --config opt-125m --batch 4 --ref-batch 0 --steps 30 --warmup 8 --no-cpu-baseline > ../../$O/bisect_$c.json 2> ../../$O/bisect_$c.err)
done
timeout 300 python bench.py --config opt-125m --batch 4 --ref-batch 0 --steps 30 --warmup 8 --no-cpu-baseline > $O/bisect_HEAD.json 2> $O/bisect_HEAD.err
for c in 511867a HEAD; do
  if [ $c = HEAD ]; then d=.; else d=_bisect/$c; fi
  (cd $d && timeout 300 python bench.py --config opt-125m --batch 4 --ref-batch 0 --steps 30 --warmup 8 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bisect2_$c.json 2>/dev/null)
done
tail -3 $O/check4w.log; cat $O/time4w.log $O/time8p.log; grep -h -o '"value": [0-9.]*' $O/bisect*.json
