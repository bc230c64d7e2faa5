python tests/diag/diag_b128_bf16_loss.py 2>&1 | grep -v amdgpu.ids
SIMQ_LIBRARY=$PWD/spatial-intention-maps_amd/simq/libsimq_ablate.so SIMQ_BF16_IMG_HALF=0 python tests/diag/diag_b128_bf16_loss.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn"
