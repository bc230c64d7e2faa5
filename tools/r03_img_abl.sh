#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 0 32 1 8 16 17 64 2; do SIMQ_BF16_IMG_DBG=$d python tools/pp_check.py 2>/dev/null | grep -E "DBG" ; done | tee gpurun_out/img_abl.log
