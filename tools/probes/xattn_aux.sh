#!/bin/bash
# cache-policy bits on the streamed row loads / stores of the cross-attention kernels (tools/build_variant.py ... -DATTN_LOAD_AUX / -DATTN_STORE_AUX)
run() { for r in 1 2; do env "$@" python tools/bench_xattn.py 64 2>/dev/null | grep "^B=" | cut -c1-150 | sed "s|^|$(echo $* | sed 's|[^ ]*variants/||') : |"; done; }
run X=base
for v in lnt2 lnt16 lnt18 snt2 lsnt; do run MMGL_LIB_PATH=$PWD/variants/lib_$v.so; done
run X=base
