// Implicit-GEMM convolution for gfx950 (fp32 in / fp32 accumulate on the matrix cores).
//
//   y[m][n] = sum_k A[m][k] * W[n][k]      m = (b, oy, ox)  n = cout  k = (ky, kx, ci)
//
// A is gathered on the fly from the NHWC activation (zero padding), W is the OHWI
// weight, i.e. both operands are K-contiguous.  This one kernel serves
//   * every forward convolution of FCN.forward  (reference networks.py:18-26,
//     resnet.py:94-102: conv 7x7 s2, 3x3 s1, 1x1) and
//   * every data-gradient (dgrad) of loss.backward() (train.py:132) -- a dgrad of a
//     stride-1 convolution is the same contraction over the flipped/transposed weight
//     produced by weight_transpose_kernel.
//
// Tiling: 256 threads = 4 waves (WM x WN); block tile BM x BN, K-step 16; each wave owns
// (BM/WM) x (BN/WN) as TM x TN tiles of v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain,
// 64 cycles/instruction/SIMD).  Operands are staged global -> VGPR -> LDS ([row][16+4 pad]
// floats, conflict-free ds_read_b128: lane (i, h) reads k = 8*kb + 4*h .. +3 of row i and
// feeds four consecutive MFMAs), double-buffered with one barrier per K-step so the next
// tile's global loads fly under the current tile's MFMAs.
//
// Epilogue (all optional, fused): +bias, per-channel sum / sum-of-squares for train-mode
// BatchNorm (fp64 atomics of per-block partials), folded-BN affine, residual add, ReLU.
#include "common.h"

namespace simq {

namespace {

constexpr int BK = 16;
constexpr int LDK = 20;  // padded row stride (floats): 80 B -> 16 distinct 16-B slots per 16 rows

struct IgemmArgs {
    const float* x;
    const float* w;
    float* y;
    const float* bias;
    double* stats;
    const float* scale;
    const float* shift;
    const float* addend;
    int relu;
    int Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    int M, K;
    int tilesN;
};

template <int BM, int BN, int WM, int WN, bool VEC>
__global__ void __launch_bounds__(256) igemm_conv_kernel(const IgemmArgs p) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must be a multiple of 32x32");
    constexpr int A_FLOATS = BM * LDK, B_FLOATS = BN * LDK;
    constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    // one LDS object: [2 stages of A|B] [row info int4 x BM]
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS + 4 * BM];
    int4* rowinfo = reinterpret_cast<int4*>(smem + 2 * STAGE_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_m = blockIdx.x / p.tilesN, tile_n = blockIdx.x % p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- row -> pixel decode, once per block (rows do not change along K) ----
    for (int r = tid; r < BM; r += 256) {
        int m = m0 + r;
        int4 ri = make_int4(0, 0, 0, 0);
        if (m < p.M) {
            int hw = p.Hout * p.Wout;
            int b = m / hw, rem = m - b * hw;
            int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            ri.x = b * p.Hin * p.Win;
            ri.y = oy * p.stride - p.pad;
            ri.z = ox * p.stride - p.pad;
            ri.w = 1;
        }
        rowinfo[r] = ri;
    }
    __syncthreads();

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = VEC ? (p.K / BK) : ((p.K + BK - 1) / BK);

    // ---- staging registers ----
    constexpr int A_PASSES_V = BM / 64, B_PASSES_V = (BN + 63) / 64;
    constexpr int A_PASSES_S = BM / 16, B_PASSES_S = BN / 16;
    float4 va[VEC ? A_PASSES_V : 1], vb[VEC ? B_PASSES_V : 1];
    float sa[VEC ? 1 : A_PASSES_S], sb[VEC ? 1 : B_PASSES_S];

    const int lrow = tid >> 2, kq = tid & 3;    // VEC: 4 threads x float4 per row, 64 rows per pass
    const int srow = tid >> 4, kl = tid & 15;   // SCALAR: 16 threads per row, 16 rows per pass
    int tap = 0, c0 = 0, ky = 0, kx = 0;        // VEC: position of the current K-tile

    auto load_tile = [&](int kt) {
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < A_PASSES_V; ++ps) {
                int4 ri = rowinfo[lrow + 64 * ps];
                int iy = ri.y + ky, ix = ri.z + kx;
                bool ok = ri.w && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    size_t off = (size_t)(ri.x + iy * p.Win + ix) * p.Cin + c0 + kq * 4;
                    v = *reinterpret_cast<const float4*>(p.x + off);
                }
                va[ps] = v;
            }
#pragma unroll
            for (int ps = 0; ps < B_PASSES_V; ++ps) {
                int n = lrow + 64 * ps;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (BN >= 64 || n < BN) {
                    size_t off = (size_t)(n0 + n) * p.K + (size_t)tap * p.Cin + c0 + kq * 4;
                    v = *reinterpret_cast<const float4*>(p.w + off);
                }
                vb[ps] = v;
            }
        } else {
            int k = kt * BK + kl;
            bool kok = k < p.K;
            int t = k / p.Cin, ci = k - t * p.Cin;
            int yy = t / p.S, xx = t - yy * p.S;
#pragma unroll
            for (int ps = 0; ps < A_PASSES_S; ++ps) {
                int4 ri = rowinfo[srow + 16 * ps];
                int iy = ri.y + yy, ix = ri.z + xx;
                bool ok = kok && ri.w && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                sa[ps] = ok ? p.x[(size_t)(ri.x + iy * p.Win + ix) * p.Cin + ci] : 0.f;
            }
#pragma unroll
            for (int ps = 0; ps < B_PASSES_S; ++ps) {
                int n = srow + 16 * ps;
                sb[ps] = kok ? p.w[(size_t)(n0 + n) * p.K + k] : 0.f;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * STAGE_FLOATS;
        float* Bs = As + A_FLOATS;
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < A_PASSES_V; ++ps)
                *reinterpret_cast<float4*>(As + (lrow + 64 * ps) * LDK + kq * 4) = va[ps];
#pragma unroll
            for (int ps = 0; ps < B_PASSES_V; ++ps) {
                int n = lrow + 64 * ps;
                if (BN >= 64 || n < BN) *reinterpret_cast<float4*>(Bs + n * LDK + kq * 4) = vb[ps];
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASSES_S; ++ps) As[(srow + 16 * ps) * LDK + kl] = sa[ps];
#pragma unroll
            for (int ps = 0; ps < B_PASSES_S; ++ps) Bs[(srow + 16 * ps) * LDK + kl] = sb[ps];
        }
    };
    auto advance = [&]() {   // VEC: next K-tile position
        c0 += BK;
        if (c0 >= p.Cin) {
            c0 = 0;
            ++tap;
            ++kx;
            if (kx >= p.S) { kx = 0; ++ky; }
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fi = lane & 31, fh = lane >> 5;
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) {
            if constexpr (VEC) advance();
            load_tile(kt + 1);
        }
        const float* As = smem + buf * STAGE_FLOATS;
        const float* Bs = As + A_FLOATS;
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            floatx4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const floatx4*>(As + (wm * (BM / WM) + i * 32 + fi) * LDK + kb * 8 + fh * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const floatx4*>(Bs + (wn * (BN / WN) + j * 32 + fi) * LDK + kb * 8 + fh * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue ----
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float ssum[TN], ssq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WN) + j * 32 + fi;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.scale ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m < p.M) {
                    float v = acc[i][j][r] + bias;
                    ssum[j] += v;
                    ssq[j] += v * v;
                    v = v * sc + sh;
                    const size_t o = (size_t)m * p.Cout + n;
                    if (p.addend) v += p.addend[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                }
            }
        }
    }
    if (p.stats) {   // block-uniform
        double* red = reinterpret_cast<double*>(smem);   // [WM][BN][2], reuses the (now idle) stage buffers
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = ssum[j] + __shfl_xor(ssum[j], 32);
            float q = ssq[j] + __shfl_xor(ssq[j], 32);
            if (fh == 0) {
                int c = wn * (BN / WN) + j * 32 + fi;
                red[(wm * BN + c) * 2 + 0] = (double)s;
                red[(wm * BN + c) * 2 + 1] = (double)q;
            }
        }
        __syncthreads();
        if (tid < BN) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                s += red[(i * BN + tid) * 2 + 0];
                q += red[(i * BN + tid) * 2 + 1];
            }
            unsafeAtomicAdd(p.stats + n0 + tid, s);
            unsafeAtomicAdd(p.stats + p.Cout + n0 + tid, q);
        }
    }
}

__global__ void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int taps,
                                        int cin) {
    // wt[ci][taps-1-t][co] = w[co][t][ci]; one thread per output element, co fastest (coalesced writes)
    size_t total = (size_t)cout * taps * cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int co = (int)(i % cout);
        size_t r = i / cout;
        int tf = (int)(r % taps);
        int ci = (int)(r / taps);
        wt[i] = w[((size_t)co * taps + (taps - 1 - tf)) * cin + ci];
    }
}

template <int BM, int BN, int WM, int WN, bool VEC>
int run(const IgemmArgs& a, hipStream_t stream) {
    IgemmArgs p = a;
    p.tilesN = p.Cout / BN;
    int tilesM = (p.M + BM - 1) / BM;
    dim3 grid((unsigned)(tilesM * p.tilesN));
    // algorithmic work: 2*M*N*K flops; bytes = read x once + read w once + write y once
    prof_launch_begin(0, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    hipLaunchKernelGGL((igemm_conv_kernel<BM, BN, WM, WN, VEC>), grid, dim3(256), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_conv_igemm(const float* x, const float* w, float* y, const ConvGeom& g, const ConvEpilogue& e,
                      hipStream_t stream) {
    IgemmArgs a;
    a.x = x; a.w = w; a.y = y;
    a.bias = e.bias; a.stats = e.stats; a.scale = e.scale; a.shift = e.shift; a.addend = e.addend; a.relu = e.relu;
    a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.R = g.R; a.S = g.S; a.stride = g.stride; a.pad = g.pad;
    a.M = g.M(); a.K = g.K(); a.tilesN = 0;
    SIMQ_REQUIRE(a.M > 0, "conv: empty problem");
    SIMQ_REQUIRE(g.Cout % 32 == 0, "conv_igemm: Cout=%d must be a multiple of 32", g.Cout);
    const bool vec = (g.Cin % BK) == 0;
    // fill the 256 CUs: prefer the 128-row tile only when it still yields >= 256 blocks
    const int bn = (g.Cout % 128 == 0) ? 128 : (g.Cout % 64 == 0 ? 64 : 32);
    const long blocks128 = (long)((a.M + 127) / 128) * (g.Cout / bn);
    const bool big = blocks128 >= 256;
    if (vec) {
        if (bn == 128) return big ? run<128, 128, 2, 2, true>(a, stream) : run<64, 128, 2, 2, true>(a, stream);
        if (bn == 64) return big ? run<128, 64, 2, 2, true>(a, stream) : run<64, 64, 2, 2, true>(a, stream);
        return run<128, 32, 4, 1, true>(a, stream);
    }
    SIMQ_REQUIRE(g.Cout % 64 == 0, "conv_igemm (generic gather): Cout=%d must be a multiple of 64", g.Cout);
    return big ? run<128, 64, 2, 2, false>(a, stream) : run<64, 64, 2, 2, false>(a, stream);
}

int launch_weight_transpose(const float* w, float* wt, int cout, int taps, int cin, hipStream_t stream) {
    size_t total = (size_t)cout * taps * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, stream, w, wt, cout, taps, cin);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
