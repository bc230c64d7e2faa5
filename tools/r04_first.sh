#!/bin/bash
# round 4, first GPU pass: the fused BatchNorm-1 kernels one by one, then the network-level suites that run the fused plan by default, then the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bnfuse.py -q -m gpu -s > gpurun_out/t_bnfuse.log 2>&1; echo "bnfuse rc=$?"
grep -n "passed\|failed\|Error\|error" gpurun_out/t_bnfuse.log | tail -15
python -m pytest tests/test_gpu_fcn.py tests/test_gpu_sized.py tests/test_gpu_intention.py -q -m gpu -x > gpurun_out/t_net.log 2>&1; echo "net rc=$?"
tail -5 gpurun_out/t_net.log
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_n1.json').read().strip().splitlines()[-1])
print('fp32 M2 %.1f M1 %.1f | bf16 M2 %s M1 %s' % (d['value'], d['value_fwd_bwd_only'], d['config'].get('bf16_configs2_full_step_transitions_per_s'), d['config'].get('bf16_configs2_fwd_bwd_only_transitions_per_s')))
PY
