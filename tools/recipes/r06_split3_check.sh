#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest -q -m gpu --durations=8 tests/test_gpu_ops.py -k "gemm or winograd" "tests/test_gpu_fullsize.py::test_b32_train_step_against_reference_pinned_golden" tests/test_gpu_overlap.py::test_overlapped_step_equals_the_serial_step_bit_for_bit > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/t_sel.log | tail -25
python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print('value', d['value'], 'M1', d['value_fwd_bwd_only'], 'ms', d['ms_per_step'])
r=d['roofline']; print({k:r[k] for k in ('achieved','peak','frac','avg_launch_ms','launches_per_step','kernel_ms_per_step','whole_step_executed_frac','fp32_equivalent','all_gemm_tiles_frac','wgrad_frac','traffic')})
print('bf16', d['bf16_configs2']['full_step_transitions_per_s'], d['bf16_configs2']['fwd_bwd_only_transitions_per_s'])
print('mfma leg', d.get('fp32_mfma_configs1'))
rm=d.get('roofline_fp32_mfma_configs1'); print({k:rm[k] for k in ('achieved','peak','frac','avg_launch_ms','whole_step_executed_frac')} if rm else None)
print('cpu', d['cpu_baseline']['value'], 'sustained', d['sustained']['transitions_per_s'])
PY
