#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 0 32 64 96; do SIMQ_BF16_WGRAD_IMG_DBG=$d python tools/wgrad_check.py 128 32 2>/dev/null | grep -E "l4 |l3 " | sed "s/^/DBG=$d /"; done | tee gpurun_out/wgrad_abl.log
