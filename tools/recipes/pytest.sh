#!/bin/bash
# selected GPU tests, verbose output kept:  tools/lease.sh pytest 1200 tests/test_x.py -k "expr"
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
python -m pytest -q -m gpu -s --durations=10 "$@" > gpurun_out/t_sel.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids" gpurun_out/t_sel.log | tail -150
