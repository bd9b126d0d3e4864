MMGL_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/n2_gloo.log 2>&1; echo "rc $?"; grep '^{"metric"' gpurun_out/n2_gloo.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['parallelism'], d.get('batch_sweep'), d.get('at_reference_batch',{}).get('value'), d['exchange'], 'protocol' in d, 'evaluate_loop' in d)
"; tail -3 gpurun_out/n2_gloo.log | cut -c1-200
