// Weight-gradient convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16).
//
//   dW[co][tap][ci] = sum_p dY[p][co] * X[p (+) tap][ci]        (wgrad half of loss.backward(), train.py:132)
//
// Both operands are stored [pixel][channel] (NHWC) while the contraction runs over pixels, i.e. the MFMA wants
// the STRIDED index contiguous per lane.  The tiles are staged to LDS exactly as they sit in HBM and read back
// with ds_read_b64_tr_b16 (gfx950 transpose read): in every 16-lane group lane t supplies the address of 4
// channels of pixel row (t >> 2) and receives 4 consecutive pixels of channel t -- two such reads give the
// 8 k-values of one 16x16x32 operand.  An XOR swizzle of the 32-B channel chunks by pixel-row bits keeps the
// rows that one transpose read touches on distinct banks.
// NP = 1: plain bf16; NP = 2: split-bf16 (hi*hi + hi*lo + lo*hi), see conv_igemm_bf16.hip.
// Pixel reduction split over blocks, fp32 hardware atomics into the zeroed gradient buffer.
#include <type_traits>

#include "common.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4_ __attribute__((ext_vector_type(4)));

constexpr int BRB = 32;   // pixels per reduction step (= MFMA k extent)

struct WgradBfArgs {
    const uint16_t* x[2];
    const uint16_t* dy[2];
    float* dw;
    int Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    int M, K;
    int tilesI, tilesJ, rows_per_split;
    int splits, xcd_group;   // xcd_group: the tiles of a pixel range on one XCD (see the kernel's block decomposition)
    float* slab;         // NULL, or [splits][Cout][K] partial tiles (deterministic plans, see conv_wgrad.hip)
    unsigned x_bytes, dy_bytes;
};

// swizzle of the 32-B (16-channel) chunk index by pixel row; CH = chunks per row
template <int CH>
__device__ __forceinline__ int rsw(int r) {
    if constexpr (CH >= 8) return (r & 3) | (((r >> 3) & 1) << 2);
    else if constexpr (CH == 4) return ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
    else if constexpr (CH == 2) return (r >> 3) & 1;
    else return 0;
}

// occupancy target 3 waves per SIMD: without it the compiler spreads into AGPRs (118 + 92 registers, 2 waves); measured +2 % on
// the bf16 step.  (4 or 5 waves spill; the fp32 wgrad kernel measured no gain from a 4-wave target.)
template <int TI, int TJ, int NP>
__global__ void __launch_bounds__(256, NP == 1 ? 3 : 2) wgrad_bf16_kernel(const WgradBfArgs p) {
    constexpr int NI = TI / 32, NJ = TJ / 32;
    static_assert(NI >= 1 && NJ >= 1, "tile must be a multiple of 32");
    constexpr int YROW = TI * 2, XROW = TJ * 2;                       // bytes per pixel row
    constexpr int Y_BYTES = BRB * YROW, X_BYTES = BRB * XROW;
    constexpr int PLANE_BYTES = Y_BYTES + X_BYTES;
    constexpr int STAGE_BYTES = NP * PLANE_BYTES;
    constexpr int CHY = TI / 16, CHX = TJ / 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    // Block -> (pixel range, tile).  The tilesI x tilesJ tiles of one pixel range read the SAME rows of dY and (shifted by their tap) of X.
    // Launch order (tile fastest) deals consecutive blocks to the 8 XCDs round-robin: the nine tap tiles of a 128 -> 128 layer sat on
    // eight different L2s and the launch fetched 237 MB for 38 MB of operands (PMC, B = 128) at 4 TB/s -- bound by exactly that.  With
    // xcd_group the j-th block an XCD receives (b = 8 j + k) is tile j % tiles of pixel range 8 (j / tiles) + k: the tiles of a range run
    // back to back on one XCD and re-read its rows from that L2.  The grid is rounded up to whole groups of 8 ranges; blocks past the last
    // range leave at once.
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
    const int cj_tiles = p.Cin / TJ;
    const int tap = tj / cj_tiles;
    const int cj0 = (tj - tap * cj_tiles) * TJ;
    const int ky = tap / p.S, kx = tap - ky * p.S;

    floatx4 acc[NI][NJ];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

    // loaders: 16 B (8 channels) per lane
    constexpr int Y_LANES = TI / 8, X_LANES = TJ / 8;                  // lanes per pixel row
    constexpr int Y_V = BRB * Y_LANES, X_V = BRB * X_LANES;            // 16-B vectors per tile
    constexpr int Y_PASSES = (Y_V + 255) / 256, X_PASSES = (X_V + 255) / 256;
    uint4 vy[2][NP][Y_PASSES], vx[2][NP][X_PASSES];

    // branch-free bounds-checked loads (byte offset 0xFFFFFFFF -> zero fill)
    __amdgpu_buffer_rsrc_t xr[NP], yr[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        xr[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[pl]), 0, p.x_bytes, 0x00020000);
        yr[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dy[pl]), 0, p.dy_bytes, 0x00020000);
    }
    int xb[X_PASSES], xoy[X_PASSES], xox[X_PASSES];   // pixel decomposition of the x rows this lane loads (first tile)
#pragma unroll
    for (int ps = 0; ps < X_PASSES; ++ps) {
        const int r = rbeg + (tid + 256 * ps) / X_LANES;
        xb[ps] = r / hw;
        const int rem = r - xb[ps] * hw;
        xoy[ps] = rem / p.Wout;
        xox[ps] = rem - xoy[ps] * p.Wout;
    }
    auto load_tile = [&](auto set_c, int r0) {   // r0 >= rend: everything reads zeros; calls must advance r0 by BRB
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int ps = 0; ps < Y_PASSES; ++ps) {
            const int idx = tid + 256 * ps;
            const int row = idx / Y_LANES, c8 = idx - row * Y_LANES;
            const int r = r0 + row;
            const bool ok = (Y_V % 256 == 0 || idx < Y_V) && r < rend;
            const unsigned voff = ok ? (unsigned)(r * p.Cout + i0 + c8 * 8) * 2u : 0xFFFFFFFFu;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                vy[SET][pl][ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yr[pl], voff, 0, 0));
        }
#pragma unroll
        for (int ps = 0; ps < X_PASSES; ++ps) {
            const int idx = tid + 256 * ps;
            const int row = idx / X_LANES, c8 = idx - row * X_LANES;
            const int r = r0 + row;
            // (image, oy, ox) of this lane's row are carried in registers and advanced by BRB pixels per call
            const int iy = xoy[ps] * p.stride - p.pad + ky, ix = xox[ps] * p.stride - p.pad + kx;
            const bool ok = (X_V % 256 == 0 || idx < X_V) && r < rend && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const unsigned voff = ok ? (unsigned)(((xb[ps] * p.Hin + iy) * p.Win + ix) * p.Cin + cj0 + c8 * 8) * 2u : 0xFFFFFFFFu;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                vx[SET][pl][ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr[pl], voff, 0, 0));
            xox[ps] += BRB;
            while (xox[ps] >= p.Wout) { xox[ps] -= p.Wout; ++xoy[ps]; }
            if (xoy[ps] >= p.Hout) { xoy[ps] -= p.Hout; ++xb[ps]; }
        }
    };
    auto store_tile = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        char* st = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int ps = 0; ps < Y_PASSES; ++ps) {
                const int idx = tid + 256 * ps;
                if (Y_V % 256 == 0 || idx < Y_V) {
                    const int row = idx / Y_LANES, c8 = idx - row * Y_LANES;
                    *reinterpret_cast<uint4*>(st + pl * PLANE_BYTES + row * YROW + (((c8 >> 1) ^ rsw<CHY>(row)) << 5) + ((c8 & 1) << 4)) = vy[SET][pl][ps];
                }
            }
#pragma unroll
            for (int ps = 0; ps < X_PASSES; ++ps) {
                const int idx = tid + 256 * ps;
                if (X_V % 256 == 0 || idx < X_V) {
                    const int row = idx / X_LANES, c8 = idx - row * X_LANES;
                    *reinterpret_cast<uint4*>(st + pl * PLANE_BYTES + Y_BYTES + row * XROW + (((c8 >> 1) ^ rsw<CHX>(row)) << 5) + ((c8 & 1) << 4)) = vx[SET][pl][ps];
                }
            }
        }
    };

    // transpose-read fragment: lane (t = lane & 15, g = lane >> 4) -> 8 pixels k = 8g..8g+7 of channel (chunk*16 + t)
    const int ft = lane & 15, fg = lane >> 4;
    auto frag = [&](const char* base, int rowbytes, int chunk, auto ch_c) -> bf16x8 {
        constexpr int CH = decltype(ch_c)::value;
        const int r0 = 8 * fg + (ft >> 2), r1 = r0 + 4;
        const char* a0 = base + r0 * rowbytes + ((chunk ^ rsw<CH>(r0)) << 5) + ((ft & 3) << 3);
        const char* a1 = base + r1 * rowbytes + ((chunk ^ rsw<CH>(r1)) << 5) + ((ft & 3) << 3);
        short4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_ __attribute__((address_space(3)))*)(a0));
        short4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_ __attribute__((address_space(3)))*)(a1));
        typedef short short8_ __attribute__((ext_vector_type(8)));
        short8_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](int buf) {
        const char* st = smem + buf * STAGE_BYTES;
        bf16x8 af[NP][NI], bf[NP][NJ];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
                af[pl][a] = frag(st + pl * PLANE_BYTES, YROW, wi * (TI / 32) + a, std::integral_constant<int, CHY>{});
#pragma unroll
            for (int b = 0; b < NJ; ++b)
                bf[pl][b] = frag(st + pl * PLANE_BYTES + Y_BYTES, XROW, wj * (TJ / 32) + b, std::integral_constant<int, CHX>{});
        }
        if constexpr (NP == 2) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < NJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][a], bf[0][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < NJ; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][a], bf[1][b], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < NJ; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][a], bf[0][b], acc[a][b], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if (rbeg >= rend) return;    // block-uniform
    const int nk = (rend - rbeg + BRB - 1) / BRB;
    load_tile(S0{}, rbeg);
    store_tile(S0{}, 0);
    load_tile(S1{}, rbeg + BRB);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {     // unconditional prefetch (zeros past rend): counted vmcnt waits
        load_tile(S0{}, rbeg + (kt + 2) * BRB);
        compute(0);
        store_tile(S1{}, 1);
        __syncthreads();
        load_tile(S1{}, rbeg + (kt + 3) * BRB);
        compute(1);
        store_tile(S0{}, 0);
        __syncthreads();
    }
    if (kt < nk) compute(0);

    // C/D layout: col = lane & 15 (-> ci), row = 4 * (lane >> 4) + reg (-> co)
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
        const size_t col = (size_t)tap * p.Cin + cj0 + wj * (TJ / 2) + b * 16 + ft;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * (TI / 2) + a * 16 + 4 * fg + r;
                if (p.slab) p.slab[((size_t)split * p.Cout + i) * p.K + col] = acc[a][b][r];
                else unsafeAtomicAdd(p.dw + (size_t)i * p.K + col, acc[a][b][r]);
            }
        }
    }
}

template <int TI, int TJ, int NP>
int run(const WgradBfArgs& a, hipStream_t stream, int tune_xcd_group) {
    WgradBfArgs p = a;
    p.tilesI = p.Cout / TI;
    p.tilesJ = p.R * p.S * (p.Cin / TJ);
    const int tiles = p.tilesI * p.tilesJ;
    const int rsteps = (p.M + BRB - 1) / BRB;
    int max_splits = rsteps / 4;
    if (max_splits > 96) max_splits = 96;
    if (max_splits < 1) max_splits = 1;
    int splits = 1;
    double best = 1e300;
    // block slots per CU the split count is chosen for: with a long pixel reduction the register-staged kernel is latency-bound at one
    // block per CU and two co-resident blocks cover each other's staging (bf16 configs[2], B = 128: 1 / 2 / 3 slots -> 12 213 / 12 331 /
    // 12 200 tr/s); at B = 32 the shorter reductions lose more to the extra atomics than they gain (7009 vs 6880 tr/s)
    static const int occ_env = SIMQ_TUNE_INT("SIMQ_WGRAD_BF16_OCC", 0);
    const int occ = occ_env >= 1 ? occ_env : (p.M >= 32768 ? 2 : 1);
    for (int s = 1; s <= max_splits; ++s) {
        const long rounds = ((long)tiles * s + 256 * occ - 1) / (256 * occ);
        const double cost = (double)rounds * ((rsteps + s - 1) / s + 4);
        if (cost < best * 0.999) { best = cost; splits = s; }
    }
    if (p.slab) {                                    // deterministic: every split owns a slab of Cout x K floats
        const int64_t fit = kWgradDetSlabFloats / ((int64_t)p.Cout * p.K);
        SIMQ_REQUIRE(fit >= 1, "wgrad_bf16: the deterministic slab holds %ld floats, one tile set needs %ld", (long)kWgradDetSlabFloats, (long)p.Cout * p.K);
        if (splits > fit) splits = (int)fit;
    }
    int rps = (p.M + splits - 1) / splits;
    rps = ((rps + BRB - 1) / BRB) * BRB;
    splits = (p.M + rps - 1) / rps;
    p.rows_per_split = rps;
    p.splits = splits;
    static const int group_env = SIMQ_TUNE_INT("SIMQ_WGRAD_BF16_XCD_GROUP", 1);      // (ablation build: 0 = launch order)
    p.xcd_group = (group_env && tune_xcd_group != 0 && splits >= 8 && tiles > 1) ? 1 : 0;
    const int launch_splits = p.xcd_group ? ((splits + 7) / 8) * 8 : splits;
    note_launch(NP == 2 ? "wgrad_bf16x3_reg" : "wgrad_bf16_reg");
    prof_launch_begin(1, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    hipLaunchKernelGGL((wgrad_bf16_kernel<TI, TJ, NP>), dim3((unsigned)(tiles * launch_splits)), dim3(256), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    if (p.slab) return launch_wgrad_slab_sum(p.slab, p.dw, (int64_t)p.Cout * p.K, splits, stream);
    return 0;
}

template <int NP>
int dispatch(const WgradBfArgs& a, hipStream_t stream, int tune_xcd_group) {
    if (a.Cout % 128 == 0) {
        if (a.Cin % 128 == 0) return run<128, 128, NP>(a, stream, tune_xcd_group);
        if (a.Cin % 64 == 0) return run<128, 64, NP>(a, stream, tune_xcd_group);
    }
    if (a.Cout % 64 == 0 && a.Cin % 64 == 0) return run<64, 64, NP>(a, stream, tune_xcd_group);
    if (a.Cout % 32 == 0 && a.Cin % 128 == 0) return run<32, 128, NP>(a, stream, tune_xcd_group);
    set_error("conv_wgrad_bf16: unsupported Cout=%d Cin=%d", a.Cout, a.Cin);
    return -1;
}

}  // namespace

int launch_conv_wgrad_bf16(const uint16_t* const x[2], const uint16_t* const dy[2], int nplanes, float* dw, const ConvGeom& g,
                           hipStream_t stream, float* slab, float* det_slab) {
    WgradBfArgs a;
    a.slab = det_slab;
    a.x[0] = x[0]; a.x[1] = nplanes == 2 ? x[1] : x[0];
    a.dy[0] = dy[0]; a.dy[1] = nplanes == 2 ? dy[1] : dy[0];
    a.dw = dw;
    a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.R = g.R; a.S = g.S; a.stride = g.stride; a.pad = g.pad;
    a.M = g.M(); a.K = g.K();
    a.tilesI = a.tilesJ = a.rows_per_split = 0;
    SIMQ_REQUIRE(a.M > 0, "wgrad: empty problem");
    const double xb = 2.0 * g.B * g.Hin * g.Win * g.Cin, yb = 2.0 * a.M * g.Cout;
    SIMQ_REQUIRE(xb < 4294967000.0 && yb < 4294967000.0, "conv_wgrad_bf16: tensor exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    SIMQ_REQUIRE(nplanes == 1 || nplanes == 2, "conv_wgrad_bf16: nplanes must be 1 or 2");
    if (nplanes == 1) {                                  // wide 3x3 layers: 256 x 256 ping-pong tiles (conv_wgrad_bf16_pp.hip)
        int took = try_conv_wgrad_bf16_img(x[0], dy[0], dw, g, a.x_bytes, a.dy_bytes, stream, slab);      // image tile: all nine taps per block
        if (took != 0) return took < 0 ? took : 0;
        if (!det_slab) {                                 // (its pixel split adds with fp32 atomics: not for deterministic plans)
            took = try_conv_wgrad_bf16_pp(x[0], dy[0], dw, g, a.x_bytes, a.dy_bytes, stream);
            if (took != 0) return took < 0 ? took : 0;
        }
    }
    return nplanes == 2 ? dispatch<2>(a, stream, g.tune.wgrad_xcd_group) : dispatch<1>(a, stream, g.tune.wgrad_xcd_group);
}

}  // namespace simq
