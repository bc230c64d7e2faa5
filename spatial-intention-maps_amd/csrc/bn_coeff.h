// BatchNorm coefficients computed where they are consumed (no separate finalize launch): shared by the elementwise kernels
// (elementwise.hip) and by the convolution kernels that apply a train-mode BatchNorm + ReLU while they stage their operand (common.h InBn).
// Reference: nn.BatchNorm2d forward in train / eval mode (resnet.py:24,27,57,82; networks.py:11,13).
#pragma once
#include "common.h"

namespace simq {

constexpr float BN_EPS = 1e-5f;
constexpr double BN_MOMENTUM = 0.1;

// train: mean / biased var from the conv epilogue's fp64 sum / sum-of-squares; eval: running statistics.
__device__ __forceinline__ void bn_coeff(const BnRef& b, int c, float& scale, float& shift, float& mean_f, float& invstd_f,
                                         double& mean_d, double& var_d) {
    double mean, var;
    if (b.stats) {
        mean = b.stats[c] * b.inv_rows;
        var = b.stats[b.C + c] * b.inv_rows - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)b.rmean[c];
        var = (double)b.rvar[c];
    }
    // 1/sqrt in fp64 without the (slow) fp64 sqrt / divide: fp32 rsqrt seed + two Newton steps (error < 1e-15)
    const double v = var + (double)BN_EPS;
    double invstd = (double)rsqrtf((float)v);
    invstd = invstd * (1.5 - 0.5 * v * invstd * invstd);
    invstd = invstd * (1.5 - 0.5 * v * invstd * invstd);
    scale = (float)((double)b.gamma[c] * invstd);
    shift = (float)((double)b.beta[c] - mean * (double)b.gamma[c] * invstd);
    mean_f = (float)mean; invstd_f = (float)invstd; mean_d = mean; var_d = var;
}
__device__ __forceinline__ void bn_coeff4(const BnRef& b, int c, float4& sc, float4& sh) {
    float m, i; double md, vd;
    bn_coeff(b, c, sc.x, sh.x, m, i, md, vd); bn_coeff(b, c + 1, sc.y, sh.y, m, i, md, vd);
    bn_coeff(b, c + 2, sc.z, sh.z, m, i, md, vd); bn_coeff(b, c + 3, sc.w, sh.w, m, i, md, vd);
}
// scale / shift of a train- or eval-mode BatchNorm for every channel, once per block (fp64 mean / variance / rsqrt per channel:
// per thread it cost more than the short grid-stride loops that follow).  cs: [2][kCoeffMaxC] floats of LDS
constexpr int kCoeffMaxC = 512;
__device__ __forceinline__ void bn_coeff_block(const BnRef& b, float* cs) {
    for (int c = threadIdx.x; c < b.C; c += blockDim.x) {
        float sc, sh, m, i; double md, vd;
        bn_coeff(b, c, sc, sh, m, i, md, vd);
        cs[c] = sc; cs[kCoeffMaxC + c] = sh;
    }
}
// once per launch (block 0): save mean / invstd for backward and update the running statistics (momentum 0.1,
// unbiased variance, in fp64 like ATen's CPU kernel)
// Block (0, 0, 0) only: in a 2-D grid (the batched GEMM's blockIdx.y = transform element) every row would otherwise commit -- and the
// committing block must run the whole kernel body: in the implicit GEMM's tail-split form block 0 is always a full tile (the K-sliced
// tail blocks are the LAST blocks of the grid and return before this point).
__device__ __forceinline__ void bn_commit(const BnRef& b) {
    if (!b.stats || blockIdx.x != 0 || blockIdx.y != 0 || blockIdx.z != 0) return;
    for (int c = threadIdx.x; c < b.C; c += blockDim.x) {
        float sc, sh, m, i; double md, vd;
        bn_coeff(b, c, sc, sh, m, i, md, vd);
        b.save_mean[c] = m;
        b.save_invstd[c] = i;
        if (b.save_scale) { b.save_scale[c] = sc; b.save_shift[c] = sh; }
        const double unbiased = b.rows > 1.0 ? vd * b.rows / (b.rows - 1.0) : vd;
        if (b.defer) { b.defer[c] = md; b.defer[b.C + c] = unbiased; continue; }     // (applied later, in the reference's order)
        b.rmean[c] = (float)(BN_MOMENTUM * md + (1.0 - BN_MOMENTUM) * (double)b.rmean[c]);
        b.rvar[c] = (float)(BN_MOMENTUM * unbiased + (1.0 - BN_MOMENTUM) * (double)b.rvar[c]);
    }
}


// ---- the consumer side of common.h InBn ------------------------------------------------------------------------------------------
// scale / shift of the 4 channels c .. c + 3 this thread stages: formed from the producing convolution's statistics (in.live: forward
// pass) or read back as the forward pass saved them (backward pass: the SAME two numbers, so the recomputed activation is bit-identical)
__device__ __forceinline__ void inbn_coeff4(const InBn& in, int c, floatx4& sc, floatx4& sh) {
    if (in.live) {
        float4 a, b;
        bn_coeff4(in.bn, c, a, b);
        sc = floatx4{a.x, a.y, a.z, a.w}; sh = floatx4{b.x, b.y, b.z, b.w};
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { sc[k] = in.scale[c + k]; sh[k] = in.shift[c + k]; }
    }
}
// relu(v * sc + sh): the fma / max sequence of bn_apply
__device__ __forceinline__ floatx4 inbn_apply(floatx4 v, const floatx4& sc, const floatx4& sh) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fmaxf(__builtin_fmaf(v[c], sc[c], sh[c]), 0.f);
    return v;
}
// block 0 of a forward consumer commits the layer (mean / invstd / scale / shift for backward, running statistics)
__device__ __forceinline__ void inbn_commit(const InBn& in) {
    if (in.live) bn_commit(in.bn);
}

}  // namespace simq
