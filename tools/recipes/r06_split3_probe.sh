#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SIMQ_LIBRARY=$GRAFT_REPO_ROOT/spatial-intention-maps_amd/simq/libsimq_ablate.so
(for v in "128,force_bn=128" "128,force_bn=130"; do echo "== split3 variant $v"; PROBE_OPTS=gemm_split=1,force_bm=$v timeout 300 python tools/gemm_batched_probe.py 32; done) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/split3_probe_nosplit.log
