#!/bin/bash
# same-box A/B of the default bench line between the regular library and variants/lib_<name>.so:  tools/probes/ab_bench.sh <name> [bench args]
v=$1; shift
args="--no-cpu-baseline --no-protocol --no-batch-sweep --ref-batch 0 --steps 12 --warmup 3 $*"
pick() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric\"')][-1])
k=d.get('kernels',{})
print(sys.argv[2], d['value'], 'samples/s', d['ms_per_step'], 'ms/step; roofline',  (d.get('roofline') or {}).get('frac'), ';', {n:(round(v['ms_avg'],4)) for n,v in k.items() if n in ('mmgl_xattn_fwd','mmgl_xattn_bwd','mmgl_selfattn_fwd','mmgl_selfattn_bwd','mmgl_linear_fwd','mmgl_gemm_nt')})
" $1 $2; }
for r in 1 2; do
  python bench.py $args > /tmp/ab_base.log 2>&1; pick /tmp/ab_base.log base
  MMGL_LIB_PATH=$PWD/variants/lib_$v.so python bench.py $args > /tmp/ab_var.log 2>&1; pick /tmp/ab_var.log $v
done
