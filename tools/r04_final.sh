#!/bin/bash
# round 4: the whole GPU suite, then the measurements behind profiles/r04_* (tools/final_profiles.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -s --durations=12 > gpurun_out/t_all.log 2>&1; echo "suite rc=$?"
grep -n "passed\|failed\|^FAILED" gpurun_out/t_all.log | tail -8
bash tools/final_profiles.sh > gpurun_out/final_profiles.log 2>&1; echo "profiles rc=$?"
tail -c 600 gpurun_out/final_profiles.log
ls gpurun_out/final
bash tools/r04_timeline.sh > gpurun_out/timeline.log 2>&1; echo "timeline rc=$?"
