#!/bin/bash
# round 4, closing run: the driver's sequence (GPU suite, smoke()), then the measurements behind profiles/r04_* (tools/final_profiles.sh:
# bench.py, kernel traces overlapped + serial, SQ / FETCH / WRITE PMC passes for both legs) and the timelines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=12 > gpurun_out/t_all.log 2>&1; echo "suite rc=$?"
grep -n "passed\|failed\|^FAILED" gpurun_out/t_all.log | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/final_profiles.sh > gpurun_out/final_profiles.log 2>&1; echo "profiles rc=$?"
tail -c 600 gpurun_out/final_profiles.log
ls gpurun_out/final
bash tools/r04_timeline.sh > gpurun_out/timeline.log 2>&1; echo "timeline rc=$?"
