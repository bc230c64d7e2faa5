#!/bin/bash
# round 4, closing run: the driver's sequence -- build check (prebuilt libraries), GPU suite, smoke(), bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=8 > gpurun_out/t_all.log 2>&1; echo "suite rc=$?"
grep -n "passed\|failed\|^FAILED" gpurun_out/t_all.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"; tail -c 400 gpurun_out/bench_n1.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_n1.json').read().strip().splitlines()[-1])
print('fp32 M2 %.1f M1 %.1f sustained %s | bf16 M2 %s M1 %s' % (d['value'], d['value_fwd_bwd_only'], d['sustained']['transitions_per_s'], d['config'].get('bf16_configs2_full_step_transitions_per_s'), d['config'].get('bf16_configs2_fwd_bwd_only_transitions_per_s')))
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], '| cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
