#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for n in 8; do
  timeout 900 python bench.py --gpus $n --backend gloo --steps 2 --warmup 1 --no-cpu-baseline --replay 1100 > gpurun_out/bench_gloo_n$n.json 2> gpurun_out/bench_gloo_n$n.err; echo "n=$n rc=$?"
  python - <<P
import json
try:
    d=json.loads(open('gpurun_out/bench_gloo_n$n.json').read().strip().splitlines()[-1])
    print(d['n_gpus'], d['value'], d['dtype'], d['scaling'], d['config']['workload_key'], d['config']['transitions_per_step'], d['config']['gradient_transport'])
    print('   weak32:', d.get('weak32'))
    print('   per_rank:', [ (r['rank'], r['frac']) for r in d['roofline']['per_rank']])
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/bench_gloo_n$n.err').read()[-1500:])
P
done
