#!/bin/bash
# image-tile forward / dgrad kernel: per-kernel timing (ablation build), its parity tests, the configs[2] bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in ${IMG_DBGS:-0 1}; do SIMQ_LIBRARY=$PWD/spatial-intention-maps_amd/simq/libsimq_ablate.so SIMQ_BF16_IMG_DBG=$d python tools/pp_check.py 2>/dev/null | grep -E "DBG" ; done | tee gpurun_out/img_abl.log
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "pingpong or image_tile" > gpurun_out/t_i.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_i.log
python bench.py --workload configs2 --no-cpu-baseline --no-extras > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('gpurun_out/bench_c2.json').read().strip().splitlines()[-1])
print('configs2', d['value'], d['value_fwd_bwd_only'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
