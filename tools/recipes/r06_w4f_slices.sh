#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/recipes/kt.sh b32s --no-overlap --early-target 0 --plan-option fwd_overlap=0 --plan-option wgrad_overlap=0 > /dev/null
grep "wino4f_output" gpurun_out/b32s_launch_shapes.txt | head -12; grep "wino4f_output" gpurun_out/b32s_kernel_trace.txt | cut -c1-120
for rep in 1 2; do for q in 0 16 8 32; do echo -n "rep $rep slice_quads $q: "; SIMQ_W4F_OUT_SLICE_QUADS=$q python tools/ab_step.py configs1 60 2>&1 | tail -1; done; done
python -m pytest -q -m gpu tests/test_gpu_ops.py -k "winograd" 2>&1 | tail -2
