// Implicit-GEMM convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Same contraction and tiling idea as conv_igemm.hip (reference operators: every nn.Conv2d forward of
// networks.py:18-26 / resnet.py:94-102 except the 7x7 stem, and every dgrad of train.py:132), but the
// operands arrive as bf16 PLANES:
//   NP = 1  plain bf16 inputs (BASELINE configs 3 and 5)
//   NP = 2  "split-bf16": every fp32 operand v is stored as hi = bf16(v), lo = bf16(v - hi) and the product
//           is formed as hi*hi + hi*lo + lo*hi on the matrix cores (3 MFMAs, fp32 accumulate).  The dropped
//           lo*lo term and the 16-bit truncation of lo bound the relative error of each product by ~3 * 2^-18
//           (1.1e-5) -- fp32-class results at 16/3 x the fp32 MFMA rate.
// Planes are produced by the elementwise kernels that write the activation / gradient anyway
// (split_planes.hip), so the GEMM loop contains no conversion work.
//
// Block: 256 threads = 2 x 2 wave64, tile BM x BN (multiples of 32), K-step 32 (one MFMA k-extent); each wave
// owns (BM/2) x (BN/2) as TM x TN 16x16 tiles.  LDS rows are 64 B (32 bf16) with an XOR slot swizzle
// (slot ^ f(row>>2), f = {0,2,3,1}) that makes every 16-lane group of the ds_read_b128 fragment loads hit 16
// distinct 16-B slots without padding; 2 stages, one barrier per K-step, global loads two K-steps ahead.
// Output is fp32 (it feeds BatchNorm statistics / elementwise consumers); epilogue as in conv_igemm.hip.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_epilogue.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// XOR swizzle of the 16-B slots of an LDS row so that every 16-lane group of a ds_read_b128 fragment load
// (rows r..r+15 at one k-slot, two consecutive k-slots per group) hits 16 distinct slots of the 256-B bank row:
//   64-B rows (4 slots):  slot ^ {0,2,3,1}[(row >> 2) & 3]        128-B rows (8 slots):  slot ^ ((row >> 1) & 7)
template <int ROWB>
__device__ __forceinline__ int swz(int row) {
    if constexpr (ROWB == 64) return (0x1320 >> (((row >> 2) & 3) * 4)) & 3;
    else return (row >> 1) & 7;
}

// BKB = K elements per LDS stage (32 or 64); SUB = BKB / 32 MFMA k-steps per stage.
template <int BM, int BN, int NP, int BKB>
__global__ void __launch_bounds__(256) igemm_bf16_kernel(const IgemmBfArgs p) {
    static_assert(BM % 32 == 0 && BN % 32 == 0, "block tile must be a multiple of 32x32");
    static_assert(BKB == 32 || BKB == 64, "K-step");
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int ROWB = BKB * 2;                       // bytes per LDS row
    constexpr int SLOTS = ROWB / 16, SUB = BKB / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int PLANE_BYTES = A_BYTES + B_BYTES;
    constexpr int STAGE_BYTES = NP * PLANE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + 16 * BM];
    int4* rowinfo = reinterpret_cast<int4*>(smem + 2 * STAGE_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile = blockIdx.x;
    if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    const int tile_m = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

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

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BKB;
    constexpr int RPP = 256 / SLOTS;                                        // rows per loader pass
    constexpr int A_PASSES = (BM + RPP - 1) / RPP, B_PASSES = (BN + RPP - 1) / RPP;
    uint4 va[2][NP][A_PASSES], vb[2][NP][B_PASSES];
    const int lrow = tid / SLOTS, kq = tid % SLOTS;
    int tap = 0, c0 = 0, ky = 0, kx = 0;    // K order: BKB-channel chunk outer, filter tap inner (x lines re-used across taps)

    // branch-free loads: out-of-image taps / ragged rows / tiles past the end use byte offset 0xFFFFFFFF, which the
    // buffer bounds check zero-fills; rows of this thread live in registers
    __amdgpu_buffer_rsrc_t xr[NP], wr[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        xr[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[pl]), 0, p.x_bytes, 0x00020000);
        wr[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[pl]), 0, p.w_bytes, 0x00020000);
    }
    int rpix[A_PASSES], riy[A_PASSES], rix[A_PASSES];
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
        const int r = lrow + RPP * ps;
        int4 ri = (BM % RPP == 0 || r < BM) ? rowinfo[r] : make_int4(0, 0, 0, 0);
        rpix[ps] = ri.x;
        riy[ps] = ri.w ? ri.y : -(1 << 20);
        rix[ps] = ri.z;
    }
    auto load_tile = [&](auto set_c, bool live) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const int iy = riy[ps] + ky, ix = rix[ps] + kx;
            const bool ok = live && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const unsigned voff = ok ? (unsigned)((rpix[ps] + iy * p.Win + ix) * p.Cin + c0 + kq * 8) * 2u : 0xFFFFFFFFu;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                va[SET][pl][ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xr[pl], voff, 0, 0));
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            const int n = lrow + RPP * ps;
            const unsigned voff = (live && (BN % RPP == 0 || n < BN)) ? (unsigned)((n0 + n) * p.K + tap * p.Cin + c0 + kq * 8) * 2u : 0xFFFFFFFFu;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                vb[SET][pl][ps] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wr[pl], voff, 0, 0));
        }
    };
    auto store_tile = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        char* st = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const int r = lrow + RPP * ps;
                if (BM % RPP == 0 || r < BM)
                    *reinterpret_cast<uint4*>(st + pl * PLANE_BYTES + r * ROWB + ((kq ^ swz<ROWB>(r)) << 4)) = va[SET][pl][ps];
            }
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps) {
                const int n = lrow + RPP * ps;
                if (BN % RPP == 0 || n < BN)
                    *reinterpret_cast<uint4*>(st + pl * PLANE_BYTES + A_BYTES + n * ROWB + ((kq ^ swz<ROWB>(n)) << 4)) = vb[SET][pl][ps];
            }
        }
    };
    auto advance = [&]() {
        ++tap;
        ++kx;
        if (kx >= p.S) { kx = 0; ++ky; }
        if (tap >= p.R * p.S) { tap = 0; kx = 0; ky = 0; c0 += BKB; }
    };

    // v_mfma_f32_16x16x32_bf16 operands: lane l holds A[i = l & 15][k = 8*(l>>4) .. +7], B[k = 8*(l>>4) .. +7][j = l & 15]
    const int fi = lane & 15, fq = lane >> 4;
    bf16x8 af[2][NP][TM], bf[2][NP][TN];
    auto read_frags = [&](auto set_c, int buf, int sub) {          // fragments of MFMA k-step `sub` of LDS stage `buf`
        constexpr int SET = decltype(set_c)::value;
        const char* st = smem + buf * STAGE_BYTES;
        const int slot = sub * 4 + fq;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (BM / 2) + i * 16 + fi;
                af[SET][pl][i] = *reinterpret_cast<const bf16x8*>(st + pl * PLANE_BYTES + r * ROWB + ((slot ^ swz<ROWB>(r)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * (BN / 2) + j * 16 + fi;
                bf[SET][pl][j] = *reinterpret_cast<const bf16x8*>(st + pl * PLANE_BYTES + A_BYTES + r * ROWB + ((slot ^ swz<ROWB>(r)) << 4));
            }
        }
    };
    auto mfma_step = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        if constexpr (NP == 2) {   // small cross terms first, then the leading term
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[SET][1][i], bf[SET][0][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[SET][0][i], bf[SET][1][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[SET][0][i], bf[SET][0][j], acc[i][j], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // Software pipeline over MFMA k-steps (SUB per LDS stage): while k-step t runs from fragment set t & 1 the wave
    // reads the fragments of k-step t+1 (crossing into the next stage after the stage barrier), has tile k+3 in flight
    // from HBM and stores tile k+2 into the stage that was just retired.  Loads / stores are unconditional (zeros past
    // the end) so the compiler waits only for the OLDER global register set (counted vmcnt).
    // One stage: store tile (held in global set GS) for stage k+1 early, run SUB-1 inner k-steps, barrier, last k-step.
    auto stage = [&](auto fs_c, auto gs_c, int buf, bool live_next) {
        constexpr int FS = decltype(fs_c)::value;             // fragment set holding k-step 0 of this stage
        // tile k+1 (global set GS) -> other stage; then refill GS with tile k+3
        store_tile(gs_c, buf ^ 1);
        advance();
        load_tile(gs_c, live_next);
        if constexpr (SUB == 2) {
            read_frags(std::integral_constant<int, FS ^ 1>{}, buf, 1);
            mfma_step(std::integral_constant<int, FS>{});
            __syncthreads();                                  // tile k+1 visible; every wave is done reading stage `buf`
            read_frags(std::integral_constant<int, FS>{}, buf ^ 1, 0);
            mfma_step(std::integral_constant<int, FS ^ 1>{});
        } else {
            __syncthreads();
            read_frags(std::integral_constant<int, FS ^ 1>{}, buf ^ 1, 0);
            mfma_step(std::integral_constant<int, FS>{});
        }
    };

    load_tile(S0{}, true);                 // tile 0
    advance();
    load_tile(S1{}, nk > 1);               // tile 1
    store_tile(S0{}, 0);                   // tile 0 -> stage 0
    advance();
    load_tile(S0{}, nk > 2);               // tile 2
    __syncthreads();
    read_frags(S0{}, 0, 0);                // fragments of (stage 0, k-step 0)
    // global sets: S1 holds tile 1, S0 holds tile 2.  stage(k) stores tile k+1 and reloads that set with tile k+3.
    int kt = 0;
    if constexpr (SUB == 2) {
        // fragment set parity returns to 0 after every stage (2 k-steps per stage)
        for (; kt + 1 < nk; kt += 2) {
            stage(S0{}, S1{}, 0, kt + 3 < nk);
            stage(S0{}, S0{}, 1, kt + 4 < nk);
        }
        if (kt < nk) stage(S0{}, S1{}, 0, false);
    } else {
        for (; kt + 1 < nk; kt += 2) {
            stage(S0{}, S1{}, 0, kt + 3 < nk);
            stage(S1{}, S0{}, 1, kt + 4 < nk);
        }
        if (kt < nk) stage(S0{}, S1{}, 0, false);
    }
    __syncthreads();

    // ---- epilogue (igemm_epilogue.h); the stage buffers are idle now and serve as its reduction scratch ----
    igemm_epilogue<BM, BN, TM, TN>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
}

template <int BM, int BN, int NP, int BKB>
int run(const IgemmBfArgs& a, hipStream_t stream) {
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = bf16_xcd_chunk(((p.M + BM - 1) / BM) * p.tilesN, p.tilesN);
    int tilesM = (p.M + BM - 1) / BM;
    note_launch(NP == 2 ? "igemm_bf16x3_reg" : "igemm_bf16_reg");
    prof_launch_begin(2, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, NP, BKB>), dim3((unsigned)(tilesM * p.tilesN)), dim3(256), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

struct TileCfg { int bm, bn; float eff; };
constexpr TileCfg kMenu[] = {
    {128, 128, 1.00f}, {96, 128, 0.97f}, {64, 128, 0.92f}, {128, 64, 0.92f}, {96, 64, 0.88f}, {64, 64, 0.82f},
    {128, 32, 0.75f},  {96, 32, 0.72f},  {64, 32, 0.65f},  {32, 64, 0.65f},  {32, 32, 0.50f},
};

template <int NP, int BKB>
int dispatch(int bm, int bn, const IgemmBfArgs& a, hipStream_t stream) {
#define SIMQ_TILE(BM_, BN_) if (bm == BM_ && bn == BN_) return run<BM_, BN_, NP, BKB>(a, stream)
    SIMQ_TILE(128, 128); SIMQ_TILE(96, 128); SIMQ_TILE(64, 128); SIMQ_TILE(128, 64); SIMQ_TILE(96, 64);
    SIMQ_TILE(64, 64); SIMQ_TILE(128, 32); SIMQ_TILE(96, 32); SIMQ_TILE(64, 32); SIMQ_TILE(32, 64); SIMQ_TILE(32, 32);
#undef SIMQ_TILE
    set_error("conv_igemm_bf16: no kernel for tile %dx%d", bm, bn);
    return -1;
}

}  // namespace

int launch_conv_igemm_bf16(const uint16_t* const x[2], const uint16_t* const w[2], int nplanes, float* y, const ConvGeom& g,
                           const ConvEpilogue& e, hipStream_t stream) {
    IgemmBfArgs a;
    a.x[0] = x[0]; a.x[1] = nplanes == 2 ? x[1] : x[0];
    a.w[0] = w[0]; a.w[1] = nplanes == 2 ? w[1] : w[0];
    a.epi = make_epi(y, e);
    a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.R = g.R; a.S = g.S; a.stride = g.stride; a.pad = g.pad;
    a.M = g.M(); a.K = g.K(); a.tilesN = 0;
    a.force_bm = g.tune.force_bm; a.force_bn = g.tune.force_bn;
    SIMQ_REQUIRE(a.M > 0, "conv: empty problem");
    const double xb = 2.0 * g.B * g.Hin * g.Win * g.Cin, wb = 2.0 * g.Cout * a.K;
    SIMQ_REQUIRE(xb < 4294967000.0 && wb < 4294967000.0, "conv_igemm_bf16: tensor exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    SIMQ_REQUIRE(nplanes == 1 || nplanes == 2, "conv_igemm_bf16: nplanes must be 1 or 2");
    SIMQ_REQUIRE(g.Cout % 32 == 0 && g.Cin % 32 == 0, "conv_igemm_bf16: Cin=%d Cout=%d must be multiples of 32", g.Cin, g.Cout);
    int bm = 0, bn = 0;
    int fbm = 0, fbn = 0;
    if (bf16_forced_tile(a, &fbm, &fbn) && g.Cout % fbn == 0) { bm = fbm; bn = fbn; }
    else {
        double best = 1e300;
        for (const TileCfg& t : kMenu) {
            if (g.Cout % t.bn != 0) continue;
            const long blocks = (long)((a.M + t.bm - 1) / t.bm) * (g.Cout / t.bn);
            const long rounds = (blocks + 255) / 256;
            double cost = (double)rounds * t.bm * t.bn / t.eff;
            if (rounds == 1) cost *= 1.25;
            if (cost < best) { best = cost; bm = t.bm; bn = t.bn; }
        }
    }
    // plain bf16: 64-deep K stages (two MFMA k-steps per barrier) whenever the channel count allows; split-bf16 keeps 32
    if (nplanes == 2) return dispatch<2, 32>(bm, bn, a, stream);
    {
        const int took = try_conv_igemm_bf16_c64(a, stream);      // 64-input-channel 3x3 layers of large batches: both operands resident in LDS
        if (took != 0) return took < 0 ? took : 0;
    }
    {
        const int took = try_conv_igemm_bf16_img(a, stream);      // 3x3 layers on the 24 x 24 maps: image-tile kernel with a halo patch in LDS
        if (took != 0) return took < 0 ? took : 0;
    }
    {
        const int took = try_conv_igemm_bf16_pp(a, stream);       // 288 x 256 ping-pong kernel (256- / 512-channel layers of large batches)
        if (took != 0) return took < 0 ? took : 0;
    }
    {
        const int took = try_conv_igemm_bf16_dma(a, stream);      // large-tile LDS-DMA kernel where the shape allows
        if (took != 0) return took < 0 ? took : 0;
    }
    return (g.Cin % 64 == 0) ? dispatch<1, 64>(bm, bn, a, stream) : dispatch<1, 32>(bm, bn, a, stream);
}

}  // namespace simq
