// Weight-gradient convolution for gfx950 (fp32 matrix cores).
//
//   dW[co][ky][kx][ci] = sum_{b,oy,ox} dY[b,oy,ox,co] * X[b, oy*s-p+ky, ox*s-p+kx, ci]
//
// i.e. the wgrad half of loss.backward() (reference train.py:132) for every nn.Conv2d of
// networks.py / resnet.py.  GEMM view per filter tap: rows i = co, cols j = ci, reduction
// r = output pixel.  Both operands are read exactly as they sit in HBM (NHWC: a pixel's
// channels are contiguous, 16-B vector loads), staged to LDS as [pixel][channel] and fed to
// v_mfma_f32_32x32x2_f32 with one conflict-free ds_read_b32 per operand (lane (i,h) reads
// channel i of pixel 2*kk+h).  The pixel reduction is split over `splits` blocks per output
// tile (the output is tiny compared with the reduction) and combined with hardware fp32
// atomics into the zero-initialised gradient buffer.
#include "common.h"

namespace simq {

namespace {

constexpr int BR = 16;   // pixels per reduction step

struct WgradArgs {
    const float* x;
    const float* dy;
    float* dw;
    int Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    int M, K;
    int tilesI, tilesJ, rows_per_split;
};

template <int TI, int TJ, int WI, int WJ, bool VEC>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs p) {
    static_assert(WI * WJ == 4, "4 waves per block");
    constexpr int NI = TI / WI / 32, NJ = TJ / WJ / 32;
    static_assert(NI >= 1 && NJ >= 1, "wave tile must be a multiple of 32x32");
    constexpr int STAGE = BR * (TI + TJ);
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WJ, wj = wave % WJ;
    int id = blockIdx.x;
    const int tj = id % p.tilesJ; id /= p.tilesJ;
    const int ti = id % p.tilesI;
    const int split = id / p.tilesI;
    const int i0 = ti * TI;
    const int rbeg = split * p.rows_per_split;
    const int rend = min(p.M, rbeg + p.rows_per_split);
    const int hw = p.Hout * p.Wout;

    // column (j) meaning
    int tap = 0, cj0 = 0, ky = 0, kx = 0;           // VEC: one filter tap, TJ input channels from cj0
    int sk = 0, sci = 0, sky = 0, skx = 0;          // SCALAR: this thread's column k = (sky, skx, sci)
    bool skok = false;
    if constexpr (VEC) {
        const int cj_tiles = p.Cin / TJ;
        tap = tj / cj_tiles;
        cj0 = (tj - tap * cj_tiles) * TJ;
        ky = tap / p.S;
        kx = tap - ky * p.S;
    } else {
        sk = tj * TJ + (tid % TJ);
        skok = sk < p.K;
        int t = sk / p.Cin;
        sci = sk - t * p.Cin;
        sky = t / p.S;
        skx = t - sky * p.S;
    }

    floatx16 acc[NI][NJ];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    constexpr int Y_F4 = BR * TI / 4, X_F4 = BR * TJ / 4;
    constexpr int Y_PASSES = (Y_F4 + 255) / 256, X_PASSES_V = (X_F4 + 255) / 256;
    constexpr int X_PASSES_S = BR * TJ / 256;
    float4 vy[Y_PASSES], vx[VEC ? X_PASSES_V : 1];
    float sx[VEC ? 1 : X_PASSES_S];

    auto load_tile = [&](int r0) {
#pragma unroll
        for (int ps = 0; ps < Y_PASSES; ++ps) {
            int idx = tid + 256 * ps;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (Y_F4 % 256 == 0 || idx < Y_F4) {
                int row = idx / (TI / 4), c4 = idx - row * (TI / 4);
                int r = r0 + row;
                if (r < rend) v = *reinterpret_cast<const float4*>(p.dy + (size_t)r * p.Cout + i0 + c4 * 4);
            }
            vy[ps] = v;
        }
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_V; ++ps) {
                int idx = tid + 256 * ps;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (X_F4 % 256 == 0 || idx < X_F4) {
                    int row = idx / (TJ / 4), c4 = idx - row * (TJ / 4);
                    int r = r0 + row;
                    if (r < rend) {
                        int b = r / hw, rem = r - b * hw;
                        int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                        int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                        if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
                            v = *reinterpret_cast<const float4*>(
                                p.x + ((size_t)(b * p.Hin + iy) * p.Win + ix) * p.Cin + cj0 + c4 * 4);
                    }
                }
                vx[ps] = v;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_S; ++ps) {
                int row = tid / TJ + (256 / TJ) * ps;
                int r = r0 + row;
                float v = 0.f;
                if (skok && r < rend) {
                    int b = r / hw, rem = r - b * hw;
                    int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                    int iy = oy * p.stride - p.pad + sky, ix = ox * p.stride - p.pad + skx;
                    if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)
                        v = p.x[((size_t)(b * p.Hin + iy) * p.Win + ix) * p.Cin + sci];
                }
                sx[ps] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* Ys = smem + buf * STAGE;
        float* Xs = Ys + BR * TI;
#pragma unroll
        for (int ps = 0; ps < Y_PASSES; ++ps) {
            int idx = tid + 256 * ps;
            if (Y_F4 % 256 == 0 || idx < Y_F4) *reinterpret_cast<float4*>(Ys + idx * 4) = vy[ps];
        }
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_V; ++ps) {
                int idx = tid + 256 * ps;
                if (X_F4 % 256 == 0 || idx < X_F4) *reinterpret_cast<float4*>(Xs + idx * 4) = vx[ps];
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_S; ++ps) Xs[(tid / TJ + (256 / TJ) * ps) * TJ + (tid % TJ)] = sx[ps];
        }
    };

    const int fi = lane & 31, fh = lane >> 5;
    if (rbeg < rend) {
        load_tile(rbeg);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    for (int r0 = rbeg; r0 < rend; r0 += BR) {
        const bool more = (r0 + BR) < rend;
        if (more) load_tile(r0 + BR);
        const float* Ys = smem + buf * STAGE;
        const float* Xs = Ys + BR * TI;
#pragma unroll
        for (int kk = 0; kk < BR / 2; ++kk) {
            float af[NI], bf[NJ];
#pragma unroll
            for (int a = 0; a < NI; ++a) af[a] = Ys[(kk * 2 + fh) * TI + wi * (TI / WI) + a * 32 + fi];
#pragma unroll
            for (int b = 0; b < NJ; ++b) bf[b] = Xs[(kk * 2 + fh) * TJ + wj * (TJ / WJ) + b * 32 + fi];
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < NJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (rbeg >= rend) return;

    // C/D layout: col = lane & 31 (-> ci, contiguous in memory), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (-> co)
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
        const int j = wj * (TJ / WJ) + b * 32 + fi;
        size_t col;
        bool cok = true;
        if constexpr (VEC) {
            col = (size_t)tap * p.Cin + cj0 + j;
        } else {
            col = (size_t)tj * TJ + j;
            cok = col < (size_t)p.K;
        }
        if (!cok) continue;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + wi * (TI / WI) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                unsafeAtomicAdd(p.dw + (size_t)i * p.K + col, acc[a][b][r]);
            }
        }
    }
}

template <int TI, int TJ, int WI, int WJ, bool VEC>
int run(const WgradArgs& a, hipStream_t stream) {
    WgradArgs p = a;
    p.tilesI = p.Cout / TI;
    p.tilesJ = VEC ? p.R * p.S * (p.Cin / TJ) : (p.K + TJ - 1) / TJ;
    const int tiles = p.tilesI * p.tilesJ;
    // Split the pixel reduction so that the block count fills the 256 CUs in whole rounds:
    // time ~ ceil(tiles*s / 256) * (reduction steps per block + fixed prologue/atomic-epilogue cost).
    const int rsteps = (p.M + BR - 1) / BR;
    int max_splits = rsteps / 8;                     // keep >= 8 reduction steps per block
    if (max_splits > 96) max_splits = 96;
    if (max_splits < 1) max_splits = 1;
    int splits = 1;
    double best = 1e300;
    for (int s = 1; s <= max_splits; ++s) {
        const long rounds = ((long)tiles * s + 255) / 256;
        const double cost = (double)rounds * ((rsteps + s - 1) / s + 6);
        if (cost < best * 0.999) { best = cost; splits = s; }
    }
    int rps = (p.M + splits - 1) / splits;
    rps = ((rps + BR - 1) / BR) * BR;
    splits = (p.M + rps - 1) / rps;
    p.rows_per_split = rps;
    prof_launch_begin(1, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WI, WJ, VEC>), dim3((unsigned)(tiles * splits)), dim3(256), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int launch_conv_wgrad(const float* x, const float* dy, float* dw, const ConvGeom& g, hipStream_t stream) {
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.R = g.R; a.S = g.S; a.stride = g.stride; a.pad = g.pad;
    a.M = g.M(); a.K = g.K();
    a.tilesI = a.tilesJ = a.rows_per_split = 0;
    SIMQ_REQUIRE(a.M > 0, "wgrad: empty problem");
    SIMQ_REQUIRE(g.Cout % 32 == 0, "conv_wgrad: Cout=%d must be a multiple of 32", g.Cout);
    const bool vec = (g.Cin % 64) == 0;
    if (vec) {
        if (g.Cout % 128 == 0) {
            if (g.Cin % 128 == 0) return run<128, 128, 2, 2, true>(a, stream);
            return run<128, 64, 2, 2, true>(a, stream);
        }
        if (g.Cout % 64 == 0) return run<64, 64, 2, 2, true>(a, stream);
        if (g.Cin % 128 == 0) return run<32, 128, 1, 4, true>(a, stream);
        SIMQ_REQUIRE(false, "conv_wgrad: unsupported Cout=%d Cin=%d", g.Cout, g.Cin);
    }
    SIMQ_REQUIRE(g.Cout % 64 == 0, "conv_wgrad (generic gather): Cout=%d must be a multiple of 64", g.Cout);
    return run<64, 64, 2, 2, false>(a, stream);
}

}  // namespace simq
