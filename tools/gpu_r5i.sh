#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
