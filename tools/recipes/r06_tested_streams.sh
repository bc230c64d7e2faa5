#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(for m in release keep; do python tools/leg_order_probe.py $m 2>&1 | grep -v amdgpu; done
echo -n "configs1                      "; bash tools/bv.sh
echo -n "configs2                      "; bash tools/bv.sh --workload configs2
echo -n "configs3 256/net groups defer "; bash tools/bv.sh --workload configs3 --steps 8 --warmup 3
echo -n "configs3 256/net sequential   "; bash tools/bv.sh --workload configs3 --steps 8 --warmup 3 --group-streams 0
echo -n "configs3 64/net groups defer  "; bash tools/bv.sh --workload configs3 --batch 64 --steps 20
echo -n "configs3 64/net sequential    "; bash tools/bv.sh --workload configs3 --batch 64 --steps 20 --group-streams 0) 2>&1 | tee gpurun_out/tested_streams.log
python -m pytest -q -m gpu tests/test_gpu_overlap.py tests/test_gpu_fcn.py -k "overlap or early or concurrent or fused_step or streams" --durations=5 2>&1 | tail -8
