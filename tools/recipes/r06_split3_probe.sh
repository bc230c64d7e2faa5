#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(for v in "128,force_bn=128" "64,force_bn=128"; do echo "== split3 tile $v"; PROBE_OPTS=gemm_split=1,force_bm=$v timeout 300 python tools/gemm_batched_probe.py 32 29; done) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/split3_probe_small.log
