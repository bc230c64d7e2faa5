#!/bin/bash
# GPU box: the per-kernel probes of round 3's last additions, written to gpurun_out/final/ (copied to profiles/r03_*):
#   imgf32_check / c64_check (the kernels of the 64-input-channel layers against the tiles they replace), the timing ablations of the
#   image-tile forward kernel (eight-wave product form and the four-wave experiment), device idle time per step (fp32, bf16)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python tools/imgf32_check.py > $O/imgf32_check.txt 2>/dev/null
python tools/c64_check.py 128 100 64 > $O/c64_check.txt 2>/dev/null
{
  echo "# tools/pp_check.py, ablation build: image-tile forward kernel, layer4 / layer3 at B = 128 (results wrong by construction)"
  echo "# SIMQ_BF16_IMG_DBG: 0 product, 1 no DMA, 8 no fragment reads, 16 no MFMAs, 17 no DMA + no MFMAs, 64 DMA issued out of range, 80 = 64 + 16, 2 no barriers"
  echo "## eight-wave ping-pong kernel (conv_igemm_bf16_img.hip)"
  for d in 0 1 8 16 17 64 80 2; do SIMQ_BF16_IMG_DBG=$d python tools/pp_check.py 2>/dev/null | grep DBG; done
  echo "## four-wave experiment (conv_igemm_bf16_img4.hip, SIMQ_BF16_IMG=2): 0 = SIMQ_BF16_IMG_DBG unset is printed by the full check below"
  for d in 1 8 16; do SIMQ_BF16_IMG=2 SIMQ_BF16_IMG_DBG=$d python tools/pp_check.py 2>/dev/null | grep DBG; done
  echo "## full check with the four-wave kernel taking the 576x128 tile (column image-tile) -- compare with the product kernel's column in a plain run"
  SIMQ_BF16_IMG=2 python tools/pp_check.py 128 2>/dev/null | grep "B="
  echo "## product (eight-wave) kernel"
  python tools/pp_check.py 128 2>/dev/null | grep "B="
} > $O/img_fwd_ablation.txt
bash tools/idle_gaps.sh fp32 --workload configs1 > /dev/null; cp gpurun_out/fp32_idle.txt $O/idle_gaps_fp32_b32.txt
bash tools/idle_gaps.sh bf16 --workload configs2 > /dev/null; cp gpurun_out/bf16_idle.txt $O/idle_gaps_bf16_b128.txt
tail -3 $O/imgf32_check.txt $O/c64_check.txt $O/idle_gaps_fp32_b32.txt
