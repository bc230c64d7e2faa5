#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( python tools/wgrad_check.py 128 64 32; SLAB=0 python tools/wgrad_check.py 128 ) > gpurun_out/wgrad_check.log 2>&1
grep wgrad gpurun_out/wgrad_check.log
python -m pytest tests/test_gpu_sized.py tests/test_gpu_ops.py -q -m gpu -x -k "sized or wgrad or config_size" > gpurun_out/t_w.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t_w.log
python bench.py --workload configs2 --no-cpu-baseline --no-extras > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/bench_c2.json').read().strip().splitlines()[-1])
print('configs2', d['value'], d['value_fwd_bwd_only'], d['roofline']['wgrad_ms_per_step'], d['roofline']['wgrad_frac'])"
