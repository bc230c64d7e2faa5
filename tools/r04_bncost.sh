#!/bin/bash
# round 4: what a BatchNorm + ReLU pass over the landed patch would cost the bf16 image-tile kernel (cost-model ablation, libsimq_ablate.so):
# whole-map form (249 of 256 registers before the pass) and half-map form (165)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for i in 1 2; do
  SIMQ_BF16_IMG_DBG=0 python tools/pp_check.py 2>&1 | grep "DBG="
  SIMQ_BF16_IMG_DBG=1024 python tools/pp_check.py 2>&1 | grep "DBG="
  SIMQ_PP_CHECK_HALF=1 SIMQ_BF16_IMG_DBG=0 python tools/pp_check.py 2>&1 | grep "DBG=" | sed 's/576x128/half-map/'
  SIMQ_PP_CHECK_HALF=1 SIMQ_BF16_IMG_DBG=1024 python tools/pp_check.py 2>&1 | grep "DBG=" | sed 's/576x128/half-map/'
done
} | tee gpurun_out/r04_img_bn_on_patch_cost_model.txt
