#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python -c "import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['batch_sweep'], d['roofline']['frac'])"
