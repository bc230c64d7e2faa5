// Kernel arguments shared by the bf16 implicit-GEMM kernels (conv_igemm_bf16.hip, conv_igemm_bf16_dma.hip).
#pragma once
#include <cstdlib>
#include "igemm_epilogue.h"

namespace simq {

struct IgemmBfArgs {
    const uint16_t* x[2];          // activation planes [pixels][Cin] (hi, lo)
    const uint16_t* w[2];          // weight planes [Cout][R*S*Cin]   (hi, lo)
    EpiArgs epi;
    int Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    int M, K;
    int tilesN;
    int xcd_chunk;                 // XCD-aware tile order (see IgemmArgs in conv_igemm.hip): blocks b < 8 * xcd_chunk take tile (b % 8) * xcd_chunk + b / 8
    unsigned x_bytes, w_bytes;     // plane sizes for the bounds-checked buffer loads
    int force_bm, force_bn;        // host side only: LaunchTune::force_bm / force_bn of the launch (0: the launchers' own rules)
};

// 1 when the launch carries a forced tile (per-kernel tests / tools; ablation build: also SIMQ_IGEMM_TILE)
inline int bf16_forced_tile(const IgemmBfArgs& a, int* bm, int* bn) {
    LaunchTune t;
    t.force_bm = a.force_bm; t.force_bn = a.force_bn;
    return tune_forced_tile(t, bm, bn);
}


// conv_igemm_bf16_dma.hip: large-tile LDS-DMA kernel for plain bf16 operands; returns 1 when it took the launch,
// 0 when the shape is not covered (the caller falls back to the register-staged kernel), < 0 on error.
int try_conv_igemm_bf16_dma(const IgemmBfArgs& a, hipStream_t stream);

// conv_igemm_bf16_pp.hip: 288 x 256 tile, wave groups one barrier apart (ping-pong); same return convention
int try_conv_igemm_bf16_pp(const IgemmBfArgs& a, hipStream_t stream);

// conv_igemm_bf16_img.hip: one 24 x 24 image x 128 output channels per block, halo patch staged once per channel chunk
int try_conv_igemm_bf16_img(const IgemmBfArgs& a, hipStream_t stream);
// conv_igemm_bf16_c64.hip: 64-input-channel 3x3 layers, half an image x 64 channels per block, both operands resident in LDS
int try_conv_igemm_bf16_c64(const IgemmBfArgs& a, hipStream_t stream);

// conv_igemm_bf16_img4.hip: the same block tile with four waves of 144 x 128 (one per SIMD, fragments double-buffered in registers)
int try_conv_igemm_bf16_img4(const IgemmBfArgs& a, hipStream_t stream);

// SIMQ_XCD_REMAP=0 keeps launch order
inline int bf16_xcd_chunk(int tiles, int tilesN) {
    static const int on = SIMQ_TUNE_INT("SIMQ_XCD_REMAP", 1) != 0 ? 1 : 0;
    return (on && tilesN > 1 && tiles >= 64) ? tiles / 8 : 0;
}

}  // namespace simq
