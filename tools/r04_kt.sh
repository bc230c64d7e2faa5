#!/bin/bash
# round 4: kernel traces of the fp32 configs[1] and bf16 configs[2] steps (7 full steps each)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/kt.sh r04_b32 > /dev/null
bash tools/kt.sh r04_bf16_b128 --workload configs2 > /dev/null
head -60 gpurun_out/r04_b32_kernel_trace.txt
