#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5v; mkdir -p $O
MMGL_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 16 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "rc $?"
tail -c 1200 $O/bench_gloo2.json; tail -5 $O/bench_gloo2.err
