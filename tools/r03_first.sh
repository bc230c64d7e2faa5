#!/bin/bash
# round 3, first GPU pass: new sized-golden tests (verbose), DP + ops tests, bench N=1 and the self-launched 2-rank form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sized.py -q -m gpu -s > gpurun_out/t_sized.log 2>&1; echo "sized rc=$?"
python -m pytest tests/test_gpu_ops.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/t_ops_dp.log 2>&1; echo "ops+dp rc=$?"
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/t_sized.log
tail -5 gpurun_out/t_ops_dp.log
