#!/bin/bash
# round 4: full GPU suite, then the single-GPU rates of the per-GPU shapes of configs[1] at N=2 (16 per GPU) and configs[3] (two nets, 64 per GPU and net)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -s -x --durations=15 > gpurun_out/t_all.log 2>&1; echo "suite rc=$?"
grep -n "passed\|failed\|^FAILED\|Error" gpurun_out/t_all.log | tail -12
grep -n "medians\|b128\|gs_b128" gpurun_out/t_all.log | tail -20
bash tools/bv.sh --workload configs1 --batch 16
bash tools/bv.sh --workload configs3 --batch 64
bash tools/bv.sh --workload configs3 --batch 256
bash tools/bv.sh --workload configs4 --batch 128
python tools/mgpu_selftest.py | tail -4
