#!/bin/bash
# full GPU suite + default bench (N=1) in one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -s -x > gpurun_out/t_all.log 2>&1; echo "suite rc=$?"
grep -n "^\[\|passed\|failed" gpurun_out/t_all.log | tail -30
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
