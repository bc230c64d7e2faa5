#!/bin/bash
# GPU box: weight gradients beside the dgrads (simq_tune_wgrad_overlap: 0 off, 1 default, 3 fp32 pairwise, 2 every precision) --
# parity tests, then alternating A-B of both bench workloads
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_bnfuse.py tests/test_gpu_fullsize.py tests/test_gpu_fcn.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do
  for on in 0 3 1; do echo -n "fp32 wgrad_overlap=$on  "; bash tools/bv.sh --wgrad-overlap $on; done
done
for on in 0 1; do
  python bench.py --no-cpu-baseline --no-extras --sustained-seconds 0 --steps 20 --warmup 5 --wgrad-overlap $on 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('wgrad_overlap=$on roofline', r['achieved'], r['frac'], r.get('avg_launch_ms'), r.get('kernel_ms_per_step'), 'M2', d['value'])"
done
echo -n "bf16 default  "; bash tools/bv.sh --workload configs2
