#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest -q -m gpu --durations=8 tests/test_gpu_bf16_points.py -k "stem_and_head" -s > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/t_sel.log | grep -v "^E  \|^    \|^$" | tail -70
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("value", d["value"], "bf16", d["bf16_configs2"]["full_step_transitions_per_s"], "third leg", (d.get("fp32_mfma_configs1") or {}).get("full_step_transitions_per_s"))'
echo -n "third leg = split again:   "; python bench.py --no-cpu-baseline --no-roofline --sustained-seconds 0 --third-leg-split 1 2>/dev/null | python -c "$P"
echo -n "headline = fp32 MFMA:      "; python bench.py --no-cpu-baseline --no-roofline --sustained-seconds 0 --plan-option gemm_split=0 2>/dev/null | python -c "$P"
