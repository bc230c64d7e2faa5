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

__device__ __forceinline__ uint16_t epi_to_bf16(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float epi_from_bf16(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

struct EpiArgs {
    float* y;
    const float* bias;
    double* stats;
    const float* scale;
    const float* shift;
    const float* addend;
    int relu;
    int y_bf16, bnr_y_bf16, addend_bf16;      // ConvEpilogue: y / bnr_y1 / bnr_y2 / addend point at bf16 values
    // fused BN-backward reduction (dgrad launches only)
    const float* bnr_mask;
    const uint16_t* bnr_mask16;
    const float* bnr_mscale; const float* bnr_mshift;   // neither mask: recomputed as bnr_y1 * mscale[n] + mshift[n] > 0
    const float* bnr_y1; const float* bnr_mean1; const float* bnr_invstd1; double* bnr_red1;
    const float* bnr_y2; const float* bnr_mean2; const float* bnr_invstd2; double* bnr_red2;
};

inline EpiArgs make_epi(float* y, const ConvEpilogue& e) {
    EpiArgs a;
    a.y = y; a.bias = e.bias; a.stats = e.stats; a.scale = e.scale; a.shift = e.shift; a.addend = e.addend; a.relu = e.relu;
    a.y_bf16 = e.y_bf16; a.bnr_y_bf16 = e.bnr_y_bf16; a.addend_bf16 = e.addend_bf16;
    a.bnr_mask = e.bnr_mask; a.bnr_mask16 = e.bnr_mask16; a.bnr_mscale = e.bnr_mscale; a.bnr_mshift = e.bnr_mshift;
    a.bnr_y1 = e.bnr_y1; a.bnr_mean1 = e.bnr_mean1; a.bnr_invstd1 = e.bnr_invstd1; a.bnr_red1 = e.bnr_red1;
    a.bnr_y2 = e.bnr_y2; a.bnr_mean2 = e.bnr_mean2; a.bnr_invstd2 = e.bnr_invstd2; a.bnr_red2 = e.bnr_red2;
    return a;
}

// smem: >= WM*BN*4 doubles, idle at this point (all waves are past the last MFMA barrier)
template <int BM, int BN, int TM, int TN, int WM = 2, int NW = 4>
__device__ __forceinline__ void igemm_epilogue(const EpiArgs& p, floatx4 (&acc)[TM][TN], int m0, int n0, int M, int Cout,
                                               void* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WN = NW / WM;                      // WM x WN waves, wave tile (BM/WM) x (BN/WN) = (16 TM) x (16 TN)
    static_assert(TM * 16 * WM == BM && TN * 16 * WN == BN, "wave layout does not cover the block tile");
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fq = lane >> 4;
    const bool bnr = p.bnr_red1 != nullptr, bnr2 = bnr && p.bnr_red2 != nullptr;
    float s0[TN], s1[TN], s2[TN], s3[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s0[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; s3[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 16 + fi;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.scale ? p.shift[n] : 0.f;
        float mu1 = 0.f, is1 = 0.f, mu2 = 0.f, is2 = 0.f, msc = 0.f, msh = 0.f;
        if (bnr) { mu1 = p.bnr_mean1[n]; is1 = p.bnr_invstd1[n]; }
        if (bnr && p.bnr_mscale) { msc = p.bnr_mscale[n]; msh = p.bnr_mshift[n]; }
        if (bnr2) { mu2 = p.bnr_mean2[n]; is2 = p.bnr_invstd2[n]; }
        // the global loads of four rows first (addend, ReLU mask, pre-BN outputs: up to 4 per element), then their arithmetic and
        // stores in the original order: element by element the loop is latency-bound (the store of one element may alias the
        // loads of the next, so the compiler keeps them in program order -- one HBM round trip per element and wave).  Four rows
        // at a time keeps the 96-register kernels (5 blocks per CU) free of spills; all TM * 4 at once did not.
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float adv[4], y1v[4], y2v[4];
            bool posv[4];
            if (p.addend || bnr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * (BM / WM) + i * 16 + 4 * fq + r;
                    const size_t o = (size_t)(m < M ? m : M - 1) * Cout + n;
                    adv[r] = !p.addend ? 0.f : p.addend_bf16 ? epi_from_bf16(reinterpret_cast<const uint16_t*>(p.addend)[o]) : p.addend[o];
                    if (bnr) {
                        y1v[r] = p.bnr_y_bf16 ? epi_from_bf16(reinterpret_cast<const uint16_t*>(p.bnr_y1)[o]) : p.bnr_y1[o];
                        posv[r] = p.bnr_mask ? p.bnr_mask[o] > 0.f : p.bnr_mask16 ? (short)p.bnr_mask16[o] > 0 : __builtin_fmaf(y1v[r], msc, msh) > 0.f;
                        if (bnr2) y2v[r] = p.bnr_y_bf16 ? epi_from_bf16(reinterpret_cast<const uint16_t*>(p.bnr_y2)[o]) : p.bnr_y2[o];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 16 + 4 * fq + r;
                if (m < M) {
                    float v = acc[i][j][r] + bias;
                    if (p.stats) { s0[j] += v; s1[j] += v * v; }
                    v = v * sc + sh;
                    const size_t o = (size_t)m * Cout + n;
                    if (p.addend) v += adv[r];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.y_bf16) reinterpret_cast<uint16_t*>(p.y)[o] = epi_to_bf16(v);
                    else p.y[o] = v;
                    if (bnr) {
                        const float dz = posv[r] ? v : 0.f;
                        s0[j] += dz;
                        s1[j] += dz * ((y1v[r] - mu1) * is1);
                        if (bnr2) { s2[j] += dz; s3[j] += dz * ((y2v[r] - mu2) * is2); }
                    }
                }
            }
        }
    }
    if (!p.stats && !bnr) return;   // block-uniform
    double* red = reinterpret_cast<double*>(smem);   // [WM wave rows][BN][4]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        float a = s0[j], b = s1[j], c = s2[j], d = s3[j];
        a += __shfl_xor(a, 16); b += __shfl_xor(b, 16); c += __shfl_xor(c, 16); d += __shfl_xor(d, 16);
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32); c += __shfl_xor(c, 32); d += __shfl_xor(d, 32);
        if (fq == 0) {
            double* q = red + ((wm * BN) + wn * (BN / WN) + j * 16 + fi) * 4;
            q[0] = (double)a; q[1] = (double)b; q[2] = (double)c; q[3] = (double)d;
        }
    }
    __syncthreads();
    if (tid < BN) {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
            a += red[(w * BN + tid) * 4 + 0]; b += red[(w * BN + tid) * 4 + 1];
            c += red[(w * BN + tid) * 4 + 2]; d += red[(w * BN + tid) * 4 + 3];
        }
        double* dst = p.stats ? p.stats : p.bnr_red1;
        unsafeAtomicAdd(dst + n0 + tid, a);
        unsafeAtomicAdd(dst + Cout + n0 + tid, b);
        if (bnr2) {
            unsafeAtomicAdd(p.bnr_red2 + n0 + tid, c);
            unsafeAtomicAdd(p.bnr_red2 + Cout + n0 + tid, d);
        }
    }
}

// ---- LDS-staged variant (conv_igemm_bf16_dma.hip) -------------------------------------------------------------------
// Same arithmetic as igemm_epilogue, but every global access is a 16-byte vector over full output rows: each wave
// transposes 48 rows of its accumulator tile at a time through a private LDS strip (ds_write_b32 by fragment layout,
// ds_read_b128 by rows), so y / addend / mask / y1 / y2 move as float4 -- 128 B (WTN = 32) or 256 B (WTN = 64) per
// output row and wave instruction -- instead of 4-byte accesses in 64-B pieces.  No block barrier until the final
// per-channel reduction: DS operations of one wave execute in order.
// RT = 16-row tiles staged per pass (divides TM).  smem: >= staged_epilogue_smem<...>() bytes, idle (all waves past the last
// MFMA barrier).
template <int BN, int TN, int WM, int NW, int RT>
constexpr int staged_epilogue_smem() { return ((NW * RT * 16 * (TN * 16 + 4) * 4 + 255) / 256) * 256 + WM * BN * 32; }

// BATCH: the all-bf16 gradient form (bf16 y / addend / mask plane / y1 / y2: the dgrads of plain-bf16 plans) issues the global
// loads of 4 row groups before it consumes any of them.  Row group by row group the loop is latency-bound: a wave has
// ~3 small loads in flight, then waits a full HBM round trip, 36 times over (measured: +140 us on a 290 us layer4 dgrad,
// unchanged when the bytes were halved).
template <int BM, int BN, int TM, int TN, int WM, int NW, int RT = 3, bool BATCH = false>
__device__ __forceinline__ void igemm_epilogue_staged(const EpiArgs& p, floatx4 (&acc)[TM][TN], int m0, int n0, int M, int Cout,
                                                      void* smem) {
    constexpr int WN = NW / WM, WTN = TN * 16;
    static_assert(TM * 16 * WM == BM && TN * 16 * WN == BN, "wave layout does not cover the block tile");
    static_assert(TM % RT == 0, "row tiles are staged RT at a time");
    constexpr int ROWS = RT * 16;
    constexpr int LDW = WTN + 4;                    // strip row stride (floats): 4*LDW mod 32 == 16 -> conflict-free writes
    constexpr int LPR = WTN / 4;                    // lanes per output row (float4 each)
    constexpr int RPI = 64 / LPR;                   // rows per wave instruction
    constexpr int STRIP = ROWS * LDW;               // floats per wave
    constexpr int RED_OFF = ((NW * STRIP * 4 + 255) / 256) * 256;
    static_assert(ROWS % RPI == 0, "rows per pass must be a multiple of the rows one wave instruction covers");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fi = lane & 15, fq = lane >> 4;
    const bool bnr = p.bnr_red1 != nullptr, bnr2 = bnr && p.bnr_red2 != nullptr;
    float* strip = reinterpret_cast<float*>(smem) + wave * STRIP;
    const int cl = (lane % LPR) * 4;                // this lane's 4 columns inside the wave tile
    const int rl = lane / LPR;
    const int n = n0 + wn * WTN + cl;
    floatx4 bias = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    floatx4 mu1 = sh, is1 = sh, mu2 = sh, is2 = sh, msc = sh, msh = sh;
    const bool mfy = bnr && !p.bnr_mask && !p.bnr_mask16;   // the mask is recomputed from the pre-BN output
    // per-channel vectors: scalar loads (parameter tensors are only 4-byte aligned inside the flat buffer)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (p.bias) bias[c] = p.bias[n + c];
        if (p.scale) { sc[c] = p.scale[n + c]; sh[c] = p.shift[n + c]; }
        if (bnr) { mu1[c] = p.bnr_mean1[n + c]; is1[c] = p.bnr_invstd1[n + c]; }
        if (mfy) { msc[c] = p.bnr_mscale[n + c]; msh[c] = p.bnr_mshift[n + c]; }
        if (bnr2) { mu2[c] = p.bnr_mean2[n + c]; is2[c] = p.bnr_invstd2[n + c]; }
    }
    floatx4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    const bool batch = BATCH && bnr && !p.stats && p.y_bf16 && (p.bnr_mask16 || mfy) && p.bnr_y_bf16 && (!p.addend || p.addend_bf16);
#pragma unroll
    for (int ig = 0; ig < TM / RT; ++ig) {
#pragma unroll
        for (int ii = 0; ii < RT; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) strip[(ii * 16 + 4 * fq + r) * LDW + j * 16 + fi] = acc[ig * RT + ii][j][r];
        if constexpr (BATCH) if (batch) {
            constexpr int NK = ROWS / RPI;
            constexpr int KB = NK % 4 == 0 ? 4 : NK % 3 == 0 ? 3 : NK % 2 == 0 ? 2 : 1;
            const uint16_t* ad16 = reinterpret_cast<const uint16_t*>(p.addend);
            const uint16_t* y116 = reinterpret_cast<const uint16_t*>(p.bnr_y1);
            const uint16_t* y216 = reinterpret_cast<const uint16_t*>(p.bnr_y2);
            uint16_t* out16 = reinterpret_cast<uint16_t*>(p.y);
            // recomputed mask: the mask load stays UNCONDITIONAL (it re-reads y1's address: the same cache lines) -- a load under a branch
            // would be waited for inside its branch and break the batch of loads in flight
            const uint16_t* mk16 = mfy ? y116 : p.bnr_mask16;
#pragma unroll
            for (int kg = 0; kg < NK / KB; ++kg) {
                ushort4 ha[KB], hm[KB], h1[KB], h2[KB];
                size_t off[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int m = m0 + wm * (TM * 16) + ig * ROWS + (kg * KB + u) * RPI + rl;
                    off[u] = (size_t)(m < M ? m : M - 1) * Cout + n;
                    if (ad16) ha[u] = *reinterpret_cast<const ushort4*>(ad16 + off[u]);
                    hm[u] = *reinterpret_cast<const ushort4*>(mk16 + off[u]);
                    h1[u] = *reinterpret_cast<const ushort4*>(y116 + off[u]);
                    if (bnr2) h2[u] = *reinterpret_cast<const ushort4*>(y216 + off[u]);
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int row = (kg * KB + u) * RPI + rl;
                    const int m = m0 + wm * (TM * 16) + ig * ROWS + row;
                    floatx4 v = *reinterpret_cast<const floatx4*>(strip + row * LDW + cl);
                    if (m < M) {
                        v += bias;
                        v = v * sc + sh;
                        if (ad16) v += floatx4{epi_from_bf16(ha[u].x), epi_from_bf16(ha[u].y), epi_from_bf16(ha[u].z), epi_from_bf16(ha[u].w)};
                        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        *reinterpret_cast<ushort4*>(out16 + off[u]) = make_ushort4(epi_to_bf16(v[0]), epi_to_bf16(v[1]), epi_to_bf16(v[2]), epi_to_bf16(v[3]));
                        floatx4 dz;
                        const floatx4 y1 = {epi_from_bf16(h1[u].x), epi_from_bf16(h1[u].y), epi_from_bf16(h1[u].z), epi_from_bf16(h1[u].w)};
                        if (mfy) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) dz[c] = __builtin_fmaf(y1[c], msc[c], msh[c]) > 0.f ? v[c] : 0.f;
                        } else {
                            dz[0] = (short)hm[u].x > 0 ? v[0] : 0.f; dz[1] = (short)hm[u].y > 0 ? v[1] : 0.f;
                            dz[2] = (short)hm[u].z > 0 ? v[2] : 0.f; dz[3] = (short)hm[u].w > 0 ? v[3] : 0.f;
                        }
                        s0 += dz;
                        s1 += dz * ((y1 - mu1) * is1);
                        if (bnr2) {
                            const floatx4 y2 = {epi_from_bf16(h2[u].x), epi_from_bf16(h2[u].y), epi_from_bf16(h2[u].z), epi_from_bf16(h2[u].w)};
                            s2 += dz;
                            s3 += dz * ((y2 - mu2) * is2);
                        }
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < ROWS / RPI; ++k) {
            const int row = k * RPI + rl;
            const int m = m0 + wm * (TM * 16) + ig * ROWS + row;
            floatx4 v = *reinterpret_cast<const floatx4*>(strip + row * LDW + cl);
            if (m < M) {
                v += bias;
                if (p.stats) { s0 += v; s1 += v * v; }
                v = v * sc + sh;
                const size_t o = (size_t)m * Cout + n;
                if (p.addend) {
                    if (p.addend_bf16) {
                        const ushort4 h = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(p.addend) + o);
                        v += floatx4{epi_from_bf16(h.x), epi_from_bf16(h.y), epi_from_bf16(h.z), epi_from_bf16(h.w)};
                    } else {
                        v += *reinterpret_cast<const floatx4*>(p.addend + o);
                    }
                }
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                if (p.y_bf16) {
                    *reinterpret_cast<ushort4*>(reinterpret_cast<uint16_t*>(p.y) + o) =
                        make_ushort4(epi_to_bf16(v[0]), epi_to_bf16(v[1]), epi_to_bf16(v[2]), epi_to_bf16(v[3]));
                } else {
                    *reinterpret_cast<floatx4*>(p.y + o) = v;
                }
                if (bnr) {
                    auto ldy = [&](const float* q) -> floatx4 {
                        if (!p.bnr_y_bf16) return *reinterpret_cast<const floatx4*>(q + o);
                        const ushort4 h = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(q) + o);
                        return floatx4{epi_from_bf16(h.x), epi_from_bf16(h.y), epi_from_bf16(h.z), epi_from_bf16(h.w)};
                    };
                    const floatx4 y1 = ldy(p.bnr_y1);
                    floatx4 dz;
                    if (p.bnr_mask) {
                        const floatx4 mk = *reinterpret_cast<const floatx4*>(p.bnr_mask + o);
#pragma unroll
                        for (int c = 0; c < 4; ++c) dz[c] = mk[c] > 0.f ? v[c] : 0.f;
                    } else if (p.bnr_mask16) {
                        const ushort4 mk = *reinterpret_cast<const ushort4*>(p.bnr_mask16 + o);
                        dz[0] = (short)mk.x > 0 ? v[0] : 0.f; dz[1] = (short)mk.y > 0 ? v[1] : 0.f;
                        dz[2] = (short)mk.z > 0 ? v[2] : 0.f; dz[3] = (short)mk.w > 0 ? v[3] : 0.f;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) dz[c] = __builtin_fmaf(y1[c], msc[c], msh[c]) > 0.f ? v[c] : 0.f;
                    }
                    s0 += dz;
                    s1 += dz * ((y1 - mu1) * is1);
                    if (bnr2) {
                        const floatx4 y2 = ldy(p.bnr_y2);
                        s2 += dz;
                        s3 += dz * ((y2 - mu2) * is2);
                    }
                }
            }
        }
    }
    if (!p.stats && !bnr) return;   // block-uniform
    double* red = reinterpret_cast<double*>(reinterpret_cast<char*>(smem) + RED_OFF);   // [WM][BN][4]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float a = s0[c], b = s1[c], cc = s2[c], d = s3[c];
#pragma unroll
        for (int x = LPR; x < 64; x <<= 1) {
            a += __shfl_xor(a, x); b += __shfl_xor(b, x); cc += __shfl_xor(cc, x); d += __shfl_xor(d, x);
        }
        if (rl == 0) {
            double* q = red + ((wm * BN) + wn * WTN + cl + c) * 4;
            q[0] = (double)a; q[1] = (double)b; q[2] = (double)cc; q[3] = (double)d;
        }
    }
    __syncthreads();
    if (tid < BN) {
        double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
            a += red[(w * BN + tid) * 4 + 0]; b += red[(w * BN + tid) * 4 + 1];
            c += red[(w * BN + tid) * 4 + 2]; d += red[(w * BN + tid) * 4 + 3];
        }
        double* dst = p.stats ? p.stats : p.bnr_red1;
        unsafeAtomicAdd(dst + n0 + tid, a);
        unsafeAtomicAdd(dst + Cout + n0 + tid, b);
        if (bnr2) {
            unsafeAtomicAdd(p.bnr_red2 + n0 + tid, c);
            unsafeAtomicAdd(p.bnr_red2 + Cout + n0 + tid, d);
        }
    }
}

}  // namespace simq
