#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k full_size > $O/pytest_full.log 2>&1; grep -n "passed\|failed\|fp32 gradients of all" $O/pytest_full.log | tail -5
for i in 1 2; do
  echo "== base" >> $O/sa.log; timeout 300 python tools/bench_selfattn.py 64 2>/dev/null >> $O/sa.log
  echo "== dkv3" >> $O/sa.log; MMGL_LIB_PATH=variants/lib_dkv3.so timeout 300 python tools/bench_selfattn.py 64 2>/dev/null >> $O/sa.log
done
cat $O/sa.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
