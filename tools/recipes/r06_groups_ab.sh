cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest -q -m gpu -x --durations=25 tests/test_gpu_overlap.py "tests/test_gpu_fullsize.py::test_b32_train_step_against_reference_pinned_golden" tests/test_gpu_fcn.py::test_library_fused_step_equals_composed_step tests/test_gpu_fcn.py::test_aliased_device_replay_buffer tests/test_gpu_intention.py > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/t_sel.log | tail -60
for rep in 1 2; do
 for spec in "seq:--group-streams 0" "defer:--group-streams 1 --group-issue defer" "stagger:--group-streams 1 --group-issue stagger"; do
  label=${spec%%:*}; args=${spec#*:}
  echo -n "rep $rep c3 b256 $label  "; bash tools/bv.sh --workload configs3 --steps 8 --warmup 3 $args
  echo -n "rep $rep c3 b64 $label  "; bash tools/bv.sh --workload configs3 --batch 64 --steps 20 $args
 done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_groups.log
