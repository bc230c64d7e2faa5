#!/bin/bash
# one python tool on the GPU box, output kept:  tools/lease.sh py 600 tools/x.py args...
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
python "$@" 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/py_$(basename "$1" .py).log | tail -120
