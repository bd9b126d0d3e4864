#!/bin/bash
# round 6: dead key tiles leave the walk (fwd, dq) and all-padding key blocks leave the dK / dV launch: bitwise against the round-5 build,
# then timings on padded batches old vs new
MMGL_LIB_PATH=$PWD/variants/lib_sa32_old.so python tools/probes/sa32_bitwise.py /tmp/sa_old.pt 2>&1 | tail -1
python tools/probes/sa32_bitwise.py /tmp/sa_new.pt /tmp/sa_old.pt 2>&1 | tail -9
for r in 1 2; do
  echo "== old"; MMGL_LIB_PATH=$PWD/variants/lib_sa32_old.so python tools/probes/sa32_padded.py 2>/dev/null | grep "^B="
  echo "== new"; python tools/probes/sa32_padded.py 2>/dev/null | grep "^B="
done
python -m pytest tests/test_selfattn_gpu.py tests/test_encoders_gpu.py -m gpu -x -q 2>&1 | tail -2
