// Weight-gradient convolution for gfx950 (fp32 matrix cores).
//
//   dW[co][ky][kx][ci] = sum_{b,oy,ox} dY[b,oy,ox,co] * X[b, oy*s-p+ky, ox*s-p+kx, ci]
//
// i.e. the wgrad half of loss.backward() (reference train.py:132) for every nn.Conv2d of
// networks.py / resnet.py.  GEMM view per filter tap: rows i = co, cols j = ci, reduction
// r = output pixel.  Both operands are read exactly as they sit in HBM (NHWC: a pixel's
// channels are contiguous, 16-B vector loads), staged to LDS as [pixel][channel] and fed to
// v_mfma_f32_32x32x2_f32 with one conflict-free ds_read_b32 per operand (lane (i,h) reads
// channel i of pixel 2*kk+h).  The pixel reduction is split over blocks (the output is tiny
// compared with the reduction) and combined with hardware fp32 atomics into the
// zero-initialised gradient buffer.  Loads are bounds-checked buffer loads (no branches, zero
// fill for padding / ragged ends) issued two reduction steps ahead of the MFMAs.
#include <cstdlib>
#include <type_traits>

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
    int splits, xcd_group;   // xcd_group: the tiles of a pixel range on one XCD (see the kernel's block decomposition)
    unsigned x_bytes, dy_bytes;
    long gx, gdy, gdw;   // batched launch (blockIdx.y = g): element offsets of the g-th x / dy / dw (conv_winograd.hip)
    float* slab;         // NULL, or [splits][Cout][K] partial tiles (deterministic plans: plain stores, summed in split order afterwards)
    const float* xscale; const float* xshift;   // XBN: x is a pre-BatchNorm output, the operand is relu(x * xscale[ci] + xshift[ci]) (common.h InBn)
};

template <int TI, int TJ, int WI, int WJ, bool VEC, bool XBN = false>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs p) {
    static_assert(WI * WJ == 4, "4 waves per block");
    static_assert(VEC || !XBN, "the BatchNorm-on-load form exists for the vector loader only");
    constexpr int NI = TI / WI / 32, NJ = TJ / WJ / 32;
    static_assert(NI >= 1 && NJ >= 1, "wave tile must be a multiple of 32x32");
    constexpr int STAGE = BR * (TI + TJ);
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave / WJ, wj = wave % WJ;
    // Block -> (pixel range, tile).  The tiles of one pixel range read the same rows of dY and (shifted by their tap) of X; in launch order
    // (tile fastest) consecutive blocks go to the 8 XCDs round-robin and every L2 fetches those rows for itself.  With xcd_group the j-th
    // block an XCD receives (b = 8 j + k) is tile j % tiles of pixel range 8 (j / tiles) + k -- the tiles of a range run back to back on one
    // XCD (conv_wgrad_bf16.hip has the measurement); the grid is rounded up to whole groups of 8 ranges, blocks past the last range leave.
    int id = blockIdx.x, split;
    {
        const int tiles = p.tilesI * p.tilesJ;
        if (p.xcd_group) {
            const int k = id & 7, j = id >> 3;
            split = (j / tiles) * 8 + k;
            id = (j + k) % tiles;                    // (staggered: the 8 XCDs do not add into the same tile of dw at the same time)
            if (split >= p.splits) return;
        } else {
            split = id / tiles;
            id -= split * tiles;
        }
    }
    const int tj = id % p.tilesJ;
    const int ti = id / p.tilesJ;
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
    float4 vy[2][Y_PASSES], vx[2][VEC ? X_PASSES_V : 1];
    float sx[2][VEC ? 1 : X_PASSES_S];
    // XBN: which of a register set's rows were real pixels (padding taps / rows past the end must stay 0 behind the BatchNorm + ReLU),
    // and this lane's four channels' coefficients (256 % (TJ / 4) == 0: the same channel quad in every pass)
    unsigned xok[2] = {0u, 0u};
    float4 xsc = make_float4(1.f, 1.f, 1.f, 1.f), xsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (XBN) {
        const int ch = cj0 + (tid % (TJ / 4)) * 4;
        xsc = make_float4(p.xscale[ch], p.xscale[ch + 1], p.xscale[ch + 2], p.xscale[ch + 3]);
        xsh = make_float4(p.xshift[ch], p.xshift[ch + 1], p.xshift[ch + 2], p.xshift[ch + 3]);
    }
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + blockIdx.y * p.gx), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + blockIdx.y * p.gdy), 0, p.dy_bytes, 0x00020000);

    // pixel decomposition of the rows this lane loads from x (VEC path), for the first tile; load_tile advances it
    int xb[VEC ? X_PASSES_V : 1], xoy[VEC ? X_PASSES_V : 1], xox[VEC ? X_PASSES_V : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int ps = 0; ps < X_PASSES_V; ++ps) {
            const int r = rbeg + (tid + 256 * ps) / (TJ / 4);
            xb[ps] = r / hw;
            const int rem = r - xb[ps] * hw;
            xoy[ps] = rem / p.Wout;
            xox[ps] = rem - xoy[ps] * p.Wout;
        }
    }
    auto load_tile = [&](auto set_c, int r0) {   // rows >= rend read zeros; calls must advance r0 by BR each time (byte offset 0xFFFFFFFF is out of bounds)
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int ps = 0; ps < Y_PASSES; ++ps) {
            const int idx = tid + 256 * ps;
            const int row = idx / (TI / 4), c4 = idx - row * (TI / 4);
            const int r = r0 + row;
            const bool ok = (Y_F4 % 256 == 0 || idx < Y_F4) && r < rend;
            const unsigned voff = ok ? (unsigned)(r * p.Cout + i0 + c4 * 4) * 4u : 0xFFFFFFFFu;
            vy[SET][ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yr, voff, 0, 0));
        }
        if constexpr (VEC) {
            if constexpr (XBN) xok[SET] = 0u;
#pragma unroll
            for (int ps = 0; ps < X_PASSES_V; ++ps) {
                const int idx = tid + 256 * ps;
                const int row = idx / (TJ / 4), c4 = idx - row * (TJ / 4);
                const int r = r0 + row;
                // (image, oy, ox) of this lane's row advance by BR pixels per call: carried in registers (xb/xoy/xox) instead
                // of two integer divisions per load
                const int iy = xoy[ps] * p.stride - p.pad + ky, ix = xox[ps] * p.stride - p.pad + kx;
                const bool ok = (X_F4 % 256 == 0 || idx < X_F4) && r < rend && (unsigned)iy < (unsigned)p.Hin &&
                                (unsigned)ix < (unsigned)p.Win;
                const unsigned voff = ok ? (unsigned)(((xb[ps] * p.Hin + iy) * p.Win + ix) * p.Cin + cj0 + c4 * 4) * 4u : 0xFFFFFFFFu;
                vx[SET][ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0));
                if constexpr (XBN) xok[SET] |= ok ? (1u << ps) : 0u;
                xox[ps] += BR;
                while (xox[ps] >= p.Wout) { xox[ps] -= p.Wout; ++xoy[ps]; }
                if (xoy[ps] >= p.Hout) { xoy[ps] -= p.Hout; ++xb[ps]; }
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_S; ++ps) {
                const int row = tid / TJ + (256 / TJ) * ps;
                const int r = r0 + row;
                const int b = r / hw, rem = r - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                const int iy = oy * p.stride - p.pad + sky, ix = ox * p.stride - p.pad + skx;
                const bool ok = skok && r < rend && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const unsigned voff = ok ? (unsigned)(((b * p.Hin + iy) * p.Win + ix) * p.Cin + sci) * 4u : 0xFFFFFFFFu;
                sx[SET][ps] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff, 0, 0));
            }
        }
    };
    auto store_tile = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        float* Ys = smem + buf * STAGE;
        float* Xs = Ys + BR * TI;
#pragma unroll
        for (int ps = 0; ps < Y_PASSES; ++ps) {
            int idx = tid + 256 * ps;
            if (Y_F4 % 256 == 0 || idx < Y_F4) *reinterpret_cast<float4*>(Ys + idx * 4) = vy[SET][ps];
        }
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_V; ++ps) {
                int idx = tid + 256 * ps;
                float4 v = vx[SET][ps];
                if constexpr (XBN) {
                    const bool ok = (xok[SET] >> ps) & 1u;
                    v.x = ok ? fmaxf(__builtin_fmaf(v.x, xsc.x, xsh.x), 0.f) : 0.f; v.y = ok ? fmaxf(__builtin_fmaf(v.y, xsc.y, xsh.y), 0.f) : 0.f;
                    v.z = ok ? fmaxf(__builtin_fmaf(v.z, xsc.z, xsh.z), 0.f) : 0.f; v.w = ok ? fmaxf(__builtin_fmaf(v.w, xsc.w, xsh.w), 0.f) : 0.f;
                }
                if (X_F4 % 256 == 0 || idx < X_F4) *reinterpret_cast<float4*>(Xs + idx * 4) = v;
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < X_PASSES_S; ++ps) Xs[(tid / TJ + (256 / TJ) * ps) * TJ + (tid % TJ)] = sx[SET][ps];
        }
    };

    const int fi = lane & 31, fh = lane >> 5;
    auto compute = [&](int buf) {
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
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if (rbeg >= rend) return;   // block-uniform
    const int nk = (rend - rbeg + BR - 1) / BR;
    load_tile(S0{}, rbeg);
    store_tile(S0{}, 0);
    load_tile(S1{}, rbeg + BR);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {   // unconditional prefetch: counted vmcnt waits, loads 2 steps ahead
        load_tile(S0{}, rbeg + (kt + 2) * BR);
        compute(0);
        store_tile(S1{}, 1);
        __syncthreads();
        load_tile(S1{}, rbeg + (kt + 3) * BR);
        compute(1);
        store_tile(S0{}, 0);
        __syncthreads();
    }
    if (kt < nk) compute(0);

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
                if (p.slab) p.slab[((size_t)split * p.Cout + i) * p.K + col] = acc[a][b][r];
                else unsafeAtomicAdd(p.dw + blockIdx.y * p.gdw + (size_t)i * p.K + col, acc[a][b][r]);
            }
        }
    }
}

template <int TI, int TJ, int WI, int WJ, bool VEC>
int run(const WgradArgs& a, hipStream_t stream, const LaunchTune& tune, int batch = 1) {
    static_assert(256 % (TJ / 4) == 0, "a lane keeps its channel quad across the passes of the x loader");
    WgradArgs p = a;
    p.tilesI = p.Cout / TI;
    p.tilesJ = VEC ? p.R * p.S * (p.Cin / TJ) : (p.K + TJ - 1) / TJ;
    const int tiles = p.tilesI * p.tilesJ;
    const int ltiles = tiles * batch;                // blocks per split over the whole (batched) launch
    // Split the pixel reduction so that the block count fills the 256 CUs in whole rounds:
    // time ~ ceil(tiles*s / 256) * (reduction steps per block + fixed prologue/atomic-epilogue cost).
    const int rsteps = (p.M + BR - 1) / BR;
    int max_splits = rsteps / 8;                     // keep >= 8 reduction steps per block
    if (max_splits > 96) max_splits = 96;
    if (max_splits < 1) max_splits = 1;
    int splits = 1;
    double best = 1e300;
    for (int s = 1; s <= max_splits; ++s) {
        // blocks per CU share the matrix pipe, so time ~ (blocks per CU) x (steps per block + fixed cost); one or two blocks
        // per CU cannot hide their own barriers / load latency (measured utilisation ~0.6 / ~0.85 of three resident blocks)
        const long rounds = ((long)ltiles * s + 255) / 256;
        const double util = rounds == 1 ? 0.6 : rounds == 2 ? 0.85 : 1.0;
        const double cost = (double)rounds * ((rsteps + s - 1) / s + 6) / util;
        if (cost < best * 0.999) { best = cost; splits = s; }
    }
    if (p.slab) {                                    // deterministic: every split owns a slab of Cout x K floats
        const int64_t fit = kWgradDetSlabFloats / ((int64_t)p.Cout * p.K);
        SIMQ_REQUIRE(batch == 1 && fit >= 1, "wgrad: the deterministic slab holds %ld floats, one tile set needs %ld", (long)kWgradDetSlabFloats, (long)p.Cout * p.K);
        if (splits > fit) splits = (int)fit;
    }
    if (const int forced_s = batch > 1 ? SIMQ_TUNE_INT("SIMQ_WGRAD_BATCHED_SPLITS", 0) : SIMQ_TUNE_INT("SIMQ_WGRAD_SPLITS", 0)) splits = forced_s;   // tuning aid (tools/wgrad_splits.py, ablation build)
    int rps = (p.M + splits - 1) / splits;
    rps = ((rps + BR - 1) / BR) * BR;
    splits = (p.M + rps - 1) / rps;
    p.rows_per_split = rps;
    p.splits = splits;
    // fp32: measured and left OFF by default (LaunchTune::wgrad_xcd_group == 2 forces it on, A-B runs): three alternating bench pairs each on two boxes, 3453 -> 3435 and
    // 3382 -> 3361 tr/s with the grouping, with or without the stagger -- the fp32 launches are short (22-72 us, 18 MB of operands at
    // B = 32) and are not bound by their L2 misses, unlike the bf16 kernel's at B = 128 (conv_wgrad_bf16.hip: 237 -> 63 MB, + 0.5-1 %)
    p.xcd_group = (tune.wgrad_xcd_group == 2 && batch == 1 && splits >= 8 && tiles > 1) ? 1 : 0;
    const int launch_splits = p.xcd_group ? ((splits + 7) / 8) * 8 : splits;
    note_launch(batch > 1 ? "wgrad_f32_batched" : "wgrad_f32");
    prof_launch_begin(1, 2.0 * p.M * p.Cout * p.K * batch,
                      4.0 * batch * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    if constexpr (VEC) {
        if (p.xscale) hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WI, WJ, VEC, true>), dim3((unsigned)(tiles * launch_splits), (unsigned)batch), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WI, WJ, VEC>), dim3((unsigned)(tiles * launch_splits), (unsigned)batch), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((wgrad_kernel<TI, TJ, WI, WJ, VEC>), dim3((unsigned)(tiles * launch_splits), (unsigned)batch), dim3(256), 0, stream, p);
    }
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    if (p.slab) return launch_wgrad_slab_sum(p.slab, p.dw, (int64_t)p.Cout * p.K, splits, stream);
    return 0;
}

__global__ void __launch_bounds__(256) wgrad_slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ dw, size_t n, int splits) {
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        float a = slab[e];
        for (int s2 = 1; s2 < splits; ++s2) a += slab[(size_t)s2 * n + e];
        dw[e] = a;
    }
}

}  // namespace

int launch_wgrad_slab_sum(const float* slab, float* dw, int64_t n, int splits, hipStream_t stream) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_slab_sum_kernel, dim3(blocks), dim3(256), 0, stream, slab, dw, (size_t)n, splits);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_conv_wgrad(const float* x, const float* dy, float* dw, const ConvGeom& g, hipStream_t stream, const InBn& in, float* det_slab) {
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.slab = det_slab;
    a.xscale = in.scale; a.xshift = in.shift;
    SIMQ_REQUIRE(!in.live, "conv_wgrad: BatchNorm-on-load takes the scale / shift the forward pass saved (not a live layer)");
    SIMQ_REQUIRE(!in.scale || g.Cin % 64 == 0, "conv_wgrad: BatchNorm-on-load needs Cin %% 64 == 0 (Cin=%d)", g.Cin);
    a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.R = g.R; a.S = g.S; a.stride = g.stride; a.pad = g.pad;
    a.M = g.M(); a.K = g.K();
    a.tilesI = a.tilesJ = a.rows_per_split = 0;
    a.gx = a.gdy = a.gdw = 0;
    SIMQ_REQUIRE(a.M > 0, "wgrad: empty problem");
    SIMQ_REQUIRE(g.Cout % 32 == 0, "conv_wgrad: Cout=%d must be a multiple of 32", g.Cout);
    const double xb = 4.0 * g.B * g.Hin * g.Win * g.Cin, yb = 4.0 * a.M * g.Cout;
    SIMQ_REQUIRE(xb < 4294967000.0 && yb < 4294967000.0, "conv_wgrad: tensor exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    const bool vec = (g.Cin % 64) == 0;
    if (vec) {
        if (g.Cout % 128 == 0) {
            // few 128x128 tiles (the 128-channel layers: 9) would need a very deep pixel split; 64x64 tiles measured 62 vs 77 us
            if (g.Cin % 128 == 0) return (g.Cout / 128) * (g.R * g.S * g.Cin / 128) >= 16 ? run<128, 128, 2, 2, true>(a, stream, g.tune)
                                                                                            : run<64, 64, 2, 2, true>(a, stream, g.tune);
            return run<128, 64, 2, 2, true>(a, stream, g.tune);
        }
        if (g.Cout % 64 == 0) return run<64, 64, 2, 2, true>(a, stream, g.tune);
        if (g.Cin % 128 == 0) return run<32, 128, 1, 4, true>(a, stream, g.tune);
        SIMQ_REQUIRE(false, "conv_wgrad: unsupported Cout=%d Cin=%d", g.Cout, g.Cin);
    }
    SIMQ_REQUIRE(g.Cout % 64 == 0, "conv_wgrad (generic gather): Cout=%d must be a multiple of 64", g.Cout);
    return run<64, 64, 2, 2, false>(a, stream, g.tune);
}

// `batch` independent contractions over the rows  dw_g[N][K] += dy_g[M][N]^T * x_g[M][K]  (row-major operands, g-th at
// base + g * rows * cols; dw zeroed by the caller) in one launch: the transform-domain weight gradients of conv_winograd.hip.
int launch_wgrad_batched(const float* x, const float* dy, float* dw, int M, int N, int K, int batch, hipStream_t stream, const LaunchTune& tune) {
    SIMQ_REQUIRE(M > 0 && N % 128 == 0 && K % 128 == 0 && batch >= 1, "wgrad_batched: M=%d N=%d K=%d batch=%d not supported", M, N, K, batch);
    WgradArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.Hin = M; a.Win = 1; a.Cin = K; a.Hout = M; a.Wout = 1; a.Cout = N; a.R = 1; a.S = 1; a.stride = 1; a.pad = 0;
    a.M = M; a.K = K;
    a.tilesI = a.tilesJ = a.rows_per_split = 0;
    a.gx = (long)M * K; a.gdy = (long)M * N; a.gdw = (long)N * K;
    a.xscale = a.xshift = nullptr; a.slab = nullptr;
    const double xb = 4.0 * M * K, yb = 4.0 * M * N;
    SIMQ_REQUIRE(xb < 4294967000.0 && yb < 4294967000.0, "wgrad_batched: operand exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    return run<128, 128, 2, 2, true>(a, stream, tune, batch);
}

}  // namespace simq
