// First convolution of the encoder (reference resnet.py:94: conv 7x7, stride 2, pad 3, Cin = 3..9 -> 64, no bias) on the bf16 matrix
// cores, for plain-bf16 plans.  The implicit-GEMM kernels need Cin % 32 == 0; with 3..9 input channels the fp32 kernel gathers its
// operands one float at a time (169 us at B = 128: 55 TF/s).  What this kernel uses instead:
//
//   * In NHWC the 7 taps x Cin channels of one filter ROW are 7*Cin CONTIGUOUS floats of x (35 at Cin = 5), starting at pixel
//     (2 oy + ky - 3, 2 ox - 3).  A row of the im2col matrix is therefore 7 contiguous runs, and the contraction becomes, per filter
//     row ky, K = 64 (7*Cin real values, zero weights behind them): two v_mfma_f32_16x16x32_bf16 k-steps.
//   * The A fragment of that MFMA is 8 consecutive k per lane = 8 consecutive floats of x: every lane loads its fragment STRAIGHT from
//     global memory (two dword-aligned global_load_dwordx4), masks the elements that fall outside the image row (left / right padding;
//     rows above / below the image are skipped per tile), converts to bf16 and feeds the matrix core.  No LDS for activations, no
//     packed copy of x.  k-groups that lie entirely behind the 7*Cin real values are not loaded (their weights are zero).
//   * The weights (64 x 7 x 64 bf16, rows padded to 456 elements: conflict-free ds_read_b128) sit in LDS for the lifetime of a block;
//     blocks are persistent (grid-stride over 16-pixel wave tiles, nine waves per block: 58 -> 42 us per launch at B = 128 against pairs of tiles on eight waves), so the weights are staged and the BatchNorm batch statistics of the
//     block (fp32 per lane -> fp64 per block) are pushed with atomics ONCE per block.
//   * Output: the pre-BatchNorm map as bf16 (like every other matrix-core convolution of a plain-bf16 plan; statistics from the fp32
//     accumulators).
#include <cstdint>

#include "common.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte load the compiler may not assume aligned

constexpr int COUT = 64, R = 7, KROW = 64;            // K per filter row (7 * Cin <= 63 real)
constexpr int WROW = R * KROW + 8;                     // LDS row of one output channel: 456 bf16 = 228 dwords (36 mod 64: conflict-free)
constexpr int SMEM_W = COUT * WROW * 2;                // 58 368 B

struct StemArgs {
    const float* x; const uint16_t* w16; uint16_t* y; double* stats;
    int B, H, W, C, Ho, Wo;
    unsigned x_bytes;
};

constexpr int NWAVE = 9;                               // waves per block (they share the LDS copy of the weights): 9 x 512 blocks = one
                                                       // 16-pixel tile per wave and trip, 4.5 waves per SIMD hide each other's load latency

__global__ void __launch_bounds__(NWAVE * 64, 2) stem_conv_bf16_kernel(const StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* wl = reinterpret_cast<uint16_t*>(smem);
    double* red = reinterpret_cast<double*>(smem + SMEM_W);          // [NWAVE][64][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // weights -> LDS (the prepared layout is already [cout][WROW]); all of a thread's loads before its first store
    {
        constexpr int NV = COUT * WROW / 8, TRIPS = (NV + NWAVE * 64 - 1) / (NWAVE * 64);
        uint4 v[TRIPS];
#pragma unroll
        for (int u = 0; u < TRIPS; ++u) {
            const int i = tid + u * NWAVE * 64;
            v[u] = i < NV ? reinterpret_cast<const uint4*>(p.w16)[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < TRIPS; ++u) {
            const int i = tid + u * NWAVE * 64;
            if (i < NV) reinterpret_cast<uint4*>(wl)[i] = v[u];
        }
    }
    __syncthreads();

    const long total = (long)p.B * p.H * p.W * p.C;                  // floats in x
    const int fi = lane & 15, kg = lane >> 4;
    const int rowf = p.W * p.C;                                      // floats per image row
    const int kreal = R * p.C;                                       // real k per filter row
    const int tiles_per_row = p.Wo / 16;
    const int ntiles = p.B * p.Ho * tiles_per_row;                   // 16-pixel tiles
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    const uint16_t* wfrag = wl + fi * WROW + kg * 8;                  // + j * 16 * WROW + ky * KROW + s * 32

    // One wave = one 16-pixel x 64-channel tile at a time (round 3; a wave used to own a PAIR of tiles with every (filter row, k-step)
    // an exposed load latency: 58 us per launch at B = 128)
    for (int tile = blockIdx.x * NWAVE + wave; tile < ntiles; tile += gridDim.x * NWAVE) {
        floatx4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const int ox0 = (tile % tiles_per_row) * 16;
        const int r = tile / tiles_per_row;
        const int oy = r % p.Ho, b = r / p.Ho;
        // per lane: pointer to the first float of its k-group at filter row 0 (may lie outside the tensor: rows above / below the image
        // are skipped before it is used); per k-step whether some lane's 8-float fragment touches the left / right padding (ballot)
        const int e_lo = (2 * (ox0 + fi) - 3) * p.C + kg * 8;
        const float* prow = p.x + ((long)(b * p.H + 2 * oy - 3) * rowf + e_lo);
        bool side[2], real[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int e = e_lo + s * 32;
            real[s] = s * 32 + kg * 8 < kreal;                       // k-groups entirely behind the 7*Cin real values are not loaded
            side[s] = __builtin_amdgcn_ballot_w64((e < 0 || e + 8 > rowf) && real[s]) != 0;
        }
#pragma unroll 1
        for (int ky = 0; ky < R; ++ky) {
            const int iy = 2 * oy + ky - 3;
            if ((unsigned)iy >= (unsigned)p.H) continue;                                      // wave-uniform: a padding row
            // the first row of the first image and the last row of the last: a fragment may reach outside the tensor
            const bool tensor_edge = (b == 0 && iy == 0) || (b == p.B - 1 && iy == p.H - 1);   // wave-uniform
            float v[2][8];
            // fragments of both k-steps: 8 consecutive floats of x each, straight from global memory (dword-aligned 16-byte loads: the
            // window starts at (2 ox - 3) * C floats), all four loads issued before the first mask
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[s][q] = 0.f;
                if (real[s]) {
                    const float* src = prow + (ky * rowf + s * 32);
                    if (!tensor_edge) {
                        const floatx4_a4 lo = *reinterpret_cast<const floatx4_a4*>(src), hi = *reinterpret_cast<const floatx4_a4*>(src + 4);
                        v[s][0] = lo[0]; v[s][1] = lo[1]; v[s][2] = lo[2]; v[s][3] = lo[3];
                        v[s][4] = hi[0]; v[s][5] = hi[1]; v[s][6] = hi[2]; v[s][7] = hi[3];
                    } else {
                        const long g0 = src - p.x;
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[s][q] = (g0 + q >= 0 && g0 + q < total) ? src[q] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (side[s]) {                                         // wave-uniform: some lane touches the left / right padding
                    const int e = e_lo + s * 32;
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[s][q] = ((unsigned)(e + q) < (unsigned)rowf) ? v[s][q] : 0.f;
                }
                bf16x8 af;
#pragma unroll
                for (int q = 0; q < 8; ++q) af[q] = (__bf16)v[s][q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wfrag + j * 16 * WROW + ky * KROW + s * 32);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[j], 0, 0, 0);
                }
            }
        }
        // ---- store (bf16) + statistics: lane holds column fi of N-tile j, rows 4 kg + r of the tile
        const size_t m0 = (size_t)tile * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float val = acc[j][r4];
                s0[j] += val; s1[j] += val * val;
                p.y[(m0 + 4 * kg + r4) * COUT + j * 16 + fi] = __builtin_bit_cast(uint16_t, (__bf16)val);
            }
    }
    if (!p.stats) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = s0[j], c = s1[j];
        a += __shfl_xor(a, 16); c += __shfl_xor(c, 16);
        a += __shfl_xor(a, 32); c += __shfl_xor(c, 32);
        if (kg == 0) { red[(wave * COUT + j * 16 + fi) * 2] = (double)a; red[(wave * COUT + j * 16 + fi) * 2 + 1] = (double)c; }
    }
    __syncthreads();
    if (tid < COUT) {
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { a += red[(w * COUT + tid) * 2]; c += red[(w * COUT + tid) * 2 + 1]; }
        unsafeAtomicAdd(p.stats + tid, a);
        unsafeAtomicAdd(p.stats + COUT + tid, c);
    }
}

// w [64][7][7][C] fp32 (OHWI) -> w16 [64][WROW] bf16: element ky * 64 + kx * C + c, zeros elsewhere
__global__ void stem_weight_prep_kernel(const float* __restrict__ w, uint16_t* __restrict__ w16, int C) {
    const int total = COUT * WROW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i / WROW, e = i - o * WROW;
        const int ky = e / KROW, n = e - ky * KROW;
        float v = 0.f;
        if (ky < R && n < R * C) v = w[((size_t)o * R + ky) * R * C + n];
        w16[i] = __builtin_bit_cast(uint16_t, (__bf16)v);
    }
}

// ---- weight gradient of the same convolution -------------------------------------------------------------------------------------
//   dW[o][ky][n] = sum over output pixels of dy[pix][o] * xrow_ky[pix][n],   n = kx * C + c < 7 C
// The contraction runs over PIXELS, so the MFMA operands need 8 consecutive pixels per lane -- strided in both tensors.  Per chunk of
// 32 output pixels a block therefore stages both operands TRANSPOSED in LDS: dyT [64 o][32 pix] from the bf16 plane of dy, and for each
// filter row XT_ky [64 n][32 pix], filled by the forward kernel's fragment gather (8 consecutive floats of x per (pixel, k-group),
// padding masked, converted) with the 8 values scattered to 8 LDS rows.  After one barrier wave ky contracts its filter row:
// 4 + 4 ds_read_b128 fragments, 16 MFMAs (64 x 64 outputs over k = 32 pixels).  Blocks are persistent; every block leaves its 64 x 7 x 7C
// partial sums in a scratch slab and a second launch adds the slabs in a fixed order (deterministic; atomics over 15 680 addresses
// from 512 blocks would be neither that nor fast).
constexpr int TROW = 40;                               // LDS row: 32 pixels + 8 pad (80 B: conflict-free ds_read_b128 over 16 rows)
constexpr int WG_SMEM = (COUT + R * KROW) * TROW * 2;  // dyT + 7 x XT = 40 960 B
constexpr int WG_WAVES = 8;

struct StemWgradArgs {
    const float* x; const uint16_t* dy; float* partial;
    int B, H, W, C, Ho, Wo;
};

__global__ void __launch_bounds__(WG_WAVES * 64, 2) stem_wgrad_bf16_kernel(const StemWgradArgs p) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[WG_SMEM / 2];
    uint16_t* dyT = lds;                                // [64][TROW]
    uint16_t* XT = lds + COUT * TROW;                   // [7][64][TROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowf = p.W * p.C, kreal = R * p.C;
    const int ngrp = (kreal + 7) / 8;                   // k-groups of a filter row that hold real values
    const long total = (long)p.B * p.H * p.W * p.C;
    const int tiles_per_row = p.Wo / 16;
    const int ntiles = p.B * p.Ho * tiles_per_row;
    const int nchunks = (ntiles + 1) / 2;
    const int fi = lane & 15, kg = lane >> 4;
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // ---- stage dyT: 32 pixels x 64 channels; lane -> (pixel, 8-channel group), 16-byte load, 8 scattered 2-byte stores
        if (tid < 256) {
            const int pix = tid >> 3, cg = tid & 7;
            const int tile = chunk * 2 + (pix >> 4);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (tile < ntiles) v = *reinterpret_cast<const uint4*>(p.dy + ((size_t)tile * 16 + (pix & 15)) * COUT + cg * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) dyT[(cg * 8 + q) * TROW + pix] = (uint16_t)(q & 1 ? w[q >> 1] >> 16 : w[q >> 1] & 0xffffu);
        }
        // ---- stage XT: slots (ky, tile, pixel, k-group); only the k-groups with real values (the others feed discarded columns)
        for (int f = tid; f < R * 2 * 16 * 8; f += WG_WAVES * 64) {
            const int grp = f & 7, i = (f >> 3) & 15, t = (f >> 7) & 1, ky = f >> 8;
            if (grp >= ngrp) continue;
            const int tile = chunk * 2 + t;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (tile < ntiles) {
                const int ox0 = (tile % tiles_per_row) * 16;
                const int r = tile / tiles_per_row;
                const int oy = r % p.Ho, b = r / p.Ho;
                const int iy = 2 * oy + ky - 3;
                if ((unsigned)iy < (unsigned)p.H) {
                    const int e = (2 * (ox0 + i) - 3) * p.C + grp * 8;                 // first float of the fragment inside the image row
                    const long g0 = (long)(b * p.H + iy) * rowf + e;
                    if (g0 >= 0 && g0 + 8 <= total) {
                        const floatx4_a4 lo = *reinterpret_cast<const floatx4_a4*>(p.x + g0), hi = *reinterpret_cast<const floatx4_a4*>(p.x + g0 + 4);
                        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = (g0 + q >= 0 && g0 + q < total) ? p.x[g0 + q] : 0.f;
                    }
                    if (e < 0 || e + 8 > rowf) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = ((unsigned)(e + q) < (unsigned)rowf) ? v[q] : 0.f;   // left / right padding
                    }
                }
            }
            uint16_t* dst = XT + (ky * KROW + grp * 8) * TROW + t * 16 + i;
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q * TROW] = __builtin_bit_cast(uint16_t, (__bf16)v[q]);
        }
        __syncthreads();
        // ---- wave ky: out[o][n] += sum over the 32 pixels of dyT[o][pix] * XT_ky[n][pix]
        if (wave < R) {
            bf16x8 af[4], bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                af[j] = *reinterpret_cast<const bf16x8*>(dyT + (j * 16 + fi) * TROW + kg * 8);
                bf[j] = *reinterpret_cast<const bf16x8*>(XT + (wave * KROW + j * 16 + fi) * TROW + kg * 8);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- this block's partial sums: slab [64 o][7 ky][7 C]; lane holds out[o = i * 16 + 4 kg + r][n = j * 16 + fi]
    if (wave < R) {
        float* slab = p.partial + (size_t)blockIdx.x * COUT * R * kreal;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = i * 16 + 4 * kg + r, n = j * 16 + fi;
                    if (n < kreal) slab[((size_t)o * R + wave) * kreal + n] = acc[i][j][r];
                }
    }
}

// dw[e] = sum over the slabs (overwrites dw): 64 outputs x 4 slab lanes per block -- lane q adds slabs q, q + 4, ... (four loads in
// flight), the four partial sums are combined in lane order: a fixed summation tree, the same bits on every run
__global__ void __launch_bounds__(256) stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n, int slabs) {
    __shared__ float sm[256];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < n) {
        int s = q;
        for (; s + 12 < slabs; s += 16) {
            a0 += partial[(size_t)s * n + e]; a1 += partial[(size_t)(s + 4) * n + e];
            a2 += partial[(size_t)(s + 8) * n + e]; a3 += partial[(size_t)(s + 12) * n + e];
        }
        for (; s < slabs; s += 4) a0 += partial[(size_t)s * n + e];
    }
    sm[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q == 0 && e < n) dw[e] = (sm[threadIdx.x] + sm[threadIdx.x + 64]) + (sm[threadIdx.x + 128] + sm[threadIdx.x + 192]);
}

}  // namespace

// slabs the weight-gradient launch will use for this batch (the caller provides slabs * 64 * 49 * C floats of scratch)
int stem_wgrad_bf16_slabs(int B, int H, int W) {
    const int nchunks = (B * (H / 2) * (W / 32) + 1) / 2;
    int blocks = (nchunks + 7) / 8;                                   // >= 8 chunks per block
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    return blocks;
}

// dy: bf16 [B][H/2][W/2][64];  dw: [64][7][7][C] fp32 (overwritten);  partial: stem_wgrad_bf16_slabs() * 64 * 49 * C floats
int launch_stem_wgrad_bf16(const float* x, const uint16_t* dy, float* dw, float* partial, int B, int H, int W, int C, hipStream_t stream) {
    SIMQ_REQUIRE(stem_conv_bf16_eligible(H, W, C, COUT, R, 2, 3), "stem_wgrad_bf16: shape %dx%dx%d not covered", H, W, C);
    SIMQ_REQUIRE(4.0 * B * H * W * C < 4294967000.0, "stem_wgrad_bf16: input too large");
    StemWgradArgs p;
    p.x = x; p.dy = dy; p.partial = partial;
    p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = H / 2; p.Wo = W / 2;
    const int slabs = stem_wgrad_bf16_slabs(B, H, W);
    note_launch("stem_wgrad_bf16");
    hipLaunchKernelGGL(stem_wgrad_bf16_kernel, dim3(slabs), dim3(WG_WAVES * 64), 0, stream, p);
    SIMQ_CHECK_LAUNCH();
    const int n = COUT * R * R * C;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, partial, dw, n, slabs);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int stem_conv_bf16_wbytes() { return COUT * WROW * 2; }

bool stem_conv_bf16_eligible(int H, int W, int C, int cout, int k, int stride, int pad) {
    return cout == COUT && k == R && stride == 2 && pad == 3 && C >= 1 && R * C <= KROW - 1 && H % 2 == 0 && W % 32 == 0;
}

int launch_stem_weight_prep(const float* w, uint16_t* w16, int C, hipStream_t stream) {
    hipLaunchKernelGGL(stem_weight_prep_kernel, dim3(32), dim3(256), 0, stream, w, w16, C);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

// y: bf16 [B][H/2][W/2][64];  stats: NULL or [2 * 64] doubles (accumulated into)
int launch_stem_conv_bf16(const float* x, const uint16_t* w16, uint16_t* y, double* stats, int B, int H, int W, int C, hipStream_t stream) {
    SIMQ_REQUIRE(stem_conv_bf16_eligible(H, W, C, COUT, R, 2, 3), "stem_conv_bf16: shape %dx%dx%d not covered", H, W, C);
    const double xb = 4.0 * B * H * W * C;
    SIMQ_REQUIRE(xb < 4294967000.0, "stem_conv_bf16: input exceeds the 4 GiB buffer-addressing limit");
    StemArgs p;
    p.x = x; p.w16 = w16; p.y = y; p.stats = stats;
    p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = H / 2; p.Wo = W / 2;
    p.x_bytes = (unsigned)xb;
    const int ntiles = B * p.Ho * (p.Wo / 16);
    int blocks = (ntiles + NWAVE - 1) / NWAVE;
    if (blocks > 512) blocks = 512;                                   // two persistent blocks per CU
    static bool attr_set = false;
    const int smem = SMEM_W + NWAVE * COUT * 2 * (int)sizeof(double);
    if (!attr_set) {
        SIMQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_conv_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    note_launch("stem_conv_bf16");
    hipLaunchKernelGGL(stem_conv_bf16_kernel, dim3(blocks), dim3(NWAVE * 64), smem, stream, p);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
