#!/bin/bash
# same-box A/B of the default bench line between two environment settings:  tools/probes/ab_env.sh VAR=a VAR=b [bench args]
A=$1; B=$2; shift 2
args="--no-cpu-baseline --no-protocol --no-batch-sweep --ref-batch 0 --steps 12 --warmup 3 $*"
pick() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric\"')][-1])
k=d.get('kernels',{})
print(sys.argv[2], d['value'], 'samples/s', d['ms_per_step'], 'ms/step; roofline', (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('us_per_launch'), 'us;', {n:(round(v['ms_avg'],4)) for n,v in k.items() if n in ('mmgl_xattn_fwd','mmgl_xattn_bwd','mmgl_selfattn_fwd','mmgl_selfattn_bwd')})
" $1 $2; }
for r in 1 2; do
  env $A python bench.py $args > /tmp/ab_a.log 2>&1; pick /tmp/ab_a.log $A
  env $B python bench.py $args > /tmp/ab_b.log 2>&1; pick /tmp/ab_b.log $B
done
