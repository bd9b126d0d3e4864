#!/bin/bash
# compile-time knobs of selfattn32.hip re-swept (ring depths, rescale threshold): B = 64, T = 640, D = 64 and the Llama shape
run() { env "$@" python tools/bench_selfattn.py 64 2>/dev/null | grep "^B=" | sed "s|^|$(echo $* | sed 's|[^ ]*variants/||') : |"; }
runl() { env "$@" python tools/probes/sa_llama.py 8 2>/dev/null | grep "^B=" | sed "s|^|$(echo $* | sed 's|[^ ]*variants/||') : |"; }
run X=base; runl X=base
for v in ns64_3 nsdq3 nsdkv3 thr4; do run MMGL_LIB_PATH=$PWD/variants/lib_$v.so; done
runl MMGL_LIB_PATH=$PWD/variants/lib_dkv128ns3.so
run X=base; runl X=base
