#!/bin/bash
# single-GPU rates of every config's PER-GPU shape (what the first multi-GPU run should show), stream overlaps on
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo -n "configs1 / 2 GPUs: 16 per GPU fp32        "; bash tools/bv.sh --batch 16
echo -n "weak32: 32 per GPU fp32                   "; bash tools/bv.sh
echo -n "configs3 / 4 GPUs: 64 per GPU and net     "; bash tools/bv.sh --workload configs3 --batch 64 --steps 20
echo -n "configs3 on one GPU: 256 per net          "; bash tools/bv.sh --workload configs3 --steps 8
echo -n "configs4 / 8 GPUs: 128 per GPU bf16       "; bash tools/bv.sh --workload configs4 --batch 128
echo -n "fp32 64 per GPU (Cin 4)                   "; bash tools/bv.sh --batch 64
echo -n "fp32 128 per GPU (Cin 4)                  "; bash tools/bv.sh --batch 128 --steps 15
echo -n "bf16 configs2 at 256 per GPU              "; bash tools/bv.sh --workload configs2 --batch 256 --steps 15
echo -n "bf16 configs2 at 64 per GPU               "; bash tools/bv.sh --workload configs2 --batch 64
