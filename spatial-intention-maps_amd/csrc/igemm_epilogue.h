// Shared epilogue of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_bf16.hip).
//
// Accumulator layout (v_mfma_*_16x16x*): acc[i][j][r] = C[row = 4*(lane>>4) + r][col = lane & 15] of the wave's
// (i, j) 16x16 tile; 2 x 2 waves per block.
//
//   v = acc (+bias[n])
//   forward, train-mode BN:   stats[n] += v, stats[Cout+n] += v*v                      (BatchNorm2d batch statistics)
//   v = v*scale[n]+shift[n] (+addend[m][n]) ; relu ; y[m][n] = v
//   dgrad, fused BN-backward reduction of the BatchNorm that FOLLOWS in the backward walk (whose output gradient
//   this launch just produced):  dz = v * (mask[m][n] > 0) ;
//        red1[n] += dz, red1[Cout+n] += dz * (y1[m][n]-mean1[n])*invstd1[n]            (-> d beta, d gamma)
//        and the same against (y2, mean2, invstd2) -> red2 for a downsample branch sharing the mask.
// Per-channel partial sums: fp32 per lane over <= 4*TM rows, then fp64 across lanes / waves / blocks (LDS + fp64 atomics).
#pragma once
#include "common.h"

namespace simq {

struct EpiArgs {
    float* y;
    const float* bias;
    double* stats;
    const float* scale;
    const float* shift;
    const float* addend;
    int relu;
    // fused BN-backward reduction (dgrad launches only)
    const float* bnr_mask;
    const float* bnr_y1; const float* bnr_mean1; const float* bnr_invstd1; double* bnr_red1;
    const float* bnr_y2; const float* bnr_mean2; const float* bnr_invstd2; double* bnr_red2;
};

inline EpiArgs make_epi(float* y, const ConvEpilogue& e) {
    EpiArgs a;
    a.y = y; a.bias = e.bias; a.stats = e.stats; a.scale = e.scale; a.shift = e.shift; a.addend = e.addend; a.relu = e.relu;
    a.bnr_mask = e.bnr_mask;
    a.bnr_y1 = e.bnr_y1; a.bnr_mean1 = e.bnr_mean1; a.bnr_invstd1 = e.bnr_invstd1; a.bnr_red1 = e.bnr_red1;
    a.bnr_y2 = e.bnr_y2; a.bnr_mean2 = e.bnr_mean2; a.bnr_invstd2 = e.bnr_invstd2; a.bnr_red2 = e.bnr_red2;
    return a;
}

// smem: >= 2*BN*4 doubles, idle at this point (all waves are past the last MFMA barrier)
template <int BM, int BN, int TM, int TN>
__device__ __forceinline__ void igemm_epilogue(const EpiArgs& p, floatx4 (&acc)[TM][TN], int m0, int n0, int M, int Cout,
                                               void* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fq = lane >> 4;
    const bool bnr = p.bnr_red1 != nullptr, bnr2 = bnr && p.bnr_red2 != nullptr;
    float s0[TN], s1[TN], s2[TN], s3[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; s3[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 16 + fi;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.scale ? p.shift[n] : 0.f;
        float mu1 = 0.f, is1 = 0.f, mu2 = 0.f, is2 = 0.f;
        if (bnr) { mu1 = p.bnr_mean1[n]; is1 = p.bnr_invstd1[n]; }
        if (bnr2) { mu2 = p.bnr_mean2[n]; is2 = p.bnr_invstd2[n]; }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / 2) + i * 16 + 4 * fq + r;
                if (m < M) {
                    float v = acc[i][j][r] + bias;
                    if (p.stats) { s0[j] += v; s1[j] += v * v; }
                    v = v * sc + sh;
                    const size_t o = (size_t)m * Cout + n;
                    if (p.addend) v += p.addend[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                    if (bnr) {
                        const float dz = p.bnr_mask[o] > 0.f ? v : 0.f;
                        s0[j] += dz;
                        s1[j] += dz * ((p.bnr_y1[o] - mu1) * is1);
                        if (bnr2) { s2[j] += dz; s3[j] += dz * ((p.bnr_y2[o] - mu2) * is2); }
                    }
                }
            }
        }
    }
    if (!p.stats && !bnr) return;   // block-uniform
    double* red = reinterpret_cast<double*>(smem);   // [2 wave rows][BN][4]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float a = s0[j], b = s1[j], c = s2[j], d = s3[j];
        a += __shfl_xor(a, 16); b += __shfl_xor(b, 16); c += __shfl_xor(c, 16); d += __shfl_xor(d, 16);
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32); c += __shfl_xor(c, 32); d += __shfl_xor(d, 32);
        if (fq == 0) {
            double* q = red + ((wm * BN) + wn * (BN / 2) + j * 16 + fi) * 4;
            q[0] = (double)a; q[1] = (double)b; q[2] = (double)c; q[3] = (double)d;
        }
    }
    __syncthreads();
    if (tid < BN) {
        const double a = red[tid * 4 + 0] + red[(BN + tid) * 4 + 0];
        const double b = red[tid * 4 + 1] + red[(BN + tid) * 4 + 1];
        double* dst = p.stats ? p.stats : p.bnr_red1;
        unsafeAtomicAdd(dst + n0 + tid, a);
        unsafeAtomicAdd(dst + Cout + n0 + tid, b);
        if (bnr2) {
            const double c = red[tid * 4 + 2] + red[(BN + tid) * 4 + 2];
            const double d = red[tid * 4 + 3] + red[(BN + tid) * 4 + 3];
            unsafeAtomicAdd(p.bnr_red2 + n0 + tid, c);
            unsafeAtomicAdd(p.bnr_red2 + Cout + n0 + tid, d);
        }
    }
}

}  // namespace simq
