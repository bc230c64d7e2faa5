#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "wgrad" 2>&1 | tail -5
python tools/wgrad_ksplit_check.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do bash tools/bv.sh; done
