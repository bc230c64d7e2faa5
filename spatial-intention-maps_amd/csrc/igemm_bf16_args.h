// Kernel arguments shared by the bf16 implicit-GEMM kernels (conv_igemm_bf16.hip, conv_igemm_bf16_dma.hip).
#pragma once
#include "igemm_epilogue.h"

namespace simq {

struct IgemmBfArgs {
    const uint16_t* x[2];          // activation planes [pixels][Cin] (hi, lo)
    const uint16_t* w[2];          // weight planes [Cout][R*S*Cin]   (hi, lo)
    EpiArgs epi;
    int Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    int M, K;
    int tilesN;
    unsigned x_bytes, w_bytes;     // plane sizes for the bounds-checked buffer loads
};


// conv_igemm_bf16_dma.hip: large-tile LDS-DMA kernel for plain bf16 operands; returns 1 when it took the launch,
// 0 when the shape is not covered (the caller falls back to the register-staged kernel), < 0 on error.
int try_conv_igemm_bf16_dma(const IgemmBfArgs& a, hipStream_t stream);

}  // namespace simq
