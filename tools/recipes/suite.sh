#!/bin/bash
# the driver's round-end sequence on one box: GPU suite, smoke(), default bench.py (both legs, roofline, cpu_baseline)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=60 "$@" > gpurun_out/t_all.log 2>&1; echo "suite rc=$?"
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/t_all.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/bench.log
