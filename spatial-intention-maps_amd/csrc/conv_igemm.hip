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
// Tiling: 256 threads = 4 wave64 as 2 x 2; block tile BM x BN (multiples of 32), K-step 16;
// each wave owns (BM/2) x (BN/2) as TM x TN tiles of v_mfma_f32_16x16x4_f32 (exact fp32
// FMA chain, 32 cycles/instruction/SIMD = the full fp32 matrix rate).  The 16-granular
// MFMA shape is what lets the host pick BM x BN per layer so that the block count is a
// whole multiple of the 256 CUs (24x24 feature maps give M = B*576 rows: at B = 32 a
// 128x128 tiling of layer4 is 576 blocks = 2.25 per CU, a 96x128 tiling is 768 = 3.0).
// Operands are staged global -> VGPR -> LDS ([row][16+8 pad] floats: every 16-lane group
// of a ds_read_b128 hits 16 distinct 16-B slots); lane (i, q) reads k = 4q..4q+3 of row i
// and feeds four consecutive MFMAs.  Two LDS stages, one barrier per K-step: the next
// tile's global loads are issued before the current tile's MFMAs and stored after them.
//
// Epilogue (all optional, fused): +bias, per-channel sum / sum-of-squares for train-mode
// BatchNorm (fp64 atomics of per-block partials), folded-BN affine, residual add, ReLU.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"
#include "bn_coeff.h"
#include "igemm_epilogue.h"

namespace simq {

namespace {

constexpr int BK = 16;
constexpr int LDK = 16;  // LDS rows are exactly one K-step (64 B); the 16-B slots of a row are XOR-swizzled by row bits so that
                         // the 16 rows x 1 slot (and the mixed-slot lane groups) of a ds_read_b128 fragment load hit 16 distinct slots
__device__ __forceinline__ int lds_sw(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }   // {0,2,3,1}[(row >> 2) & 3]

struct IgemmArgs {
    const float* x;
    const float* w;
    EpiArgs epi;
    int Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad;
    int M, K;
    int tilesN;
    unsigned x_bytes, w_bytes;   // sizes of x / w for the bounds-checked buffer loads
    // balanced last round (see run()): blocks >= full_tiles each contract `kper` K-steps of tile full_tiles + (i / splits)
    // and leave their accumulators in `partial`; igemm_tail_fixup_kernel adds the pieces and runs the epilogue
    int full_tiles, splits, kper;
    float* partial;
    // batched launch (blockIdx.y = g): g-th problem reads x + g*gx, w + g*gw and writes y + g*gy (elements); the Winograd
    // path runs its 16 transform-domain GEMMs this way (conv_winograd.hip)
    long gx, gw, gy;
    // XCD-aware tile order: block b runs on XCD b % 8 (observed dispatch order), each XCD has its own L2.  Blocks
    // b < 8 * xcd_chunk take tile (b % 8) * xcd_chunk + b / 8, so one XCD walks a CONTIGUOUS run of tiles (all N-tiles of the
    // same rows back to back) and the A rows are fetched into one L2 instead of all eight; the rest keep tile = b.
    int xcd_chunk;
    // batched launches: a block contracts `nt_run` consecutive N-tiles of its row block in ONE software pipeline (the loads of
    // the next N-tile's first K-steps are in flight while the last MFMAs of the current one run), so a short K = Cin (16-32
    // K-steps) does not pay the pipeline prologue per tile (tools/probes/shortk_probe.py); measured neutral, default 1 (see run())
    int nt_run;
    // batched launches: 1 = whole planes per XCD (see plane_xcd_map in the kernel)
    int plane_xcd;
    // XBN: x is the pre-BatchNorm output of the producing convolution; the A operand is relu(x * scale[ci] + shift[ci]) (common.h InBn)
    InBn in;
};

// occupancy target: tiles up to 96x128 run 3 blocks per CU, up to 96x64 five; the register budget is held to what that allows
// (168 / 96).  (Six waves for 96x64 compile to 78 registers without spilling but lose the software pipeline: 1445 vs 1737 tr/s;
// four for 96x128 spill.)  LDS is not the limit: 20.5 KB (96x64) / 28.7 KB (96x128) per block with unpadded swizzled rows.
template <int BM, int BN, bool VEC, bool BATCHED = false, bool XBN = false>
__global__ void __launch_bounds__(256, (BM * BN <= 96 * 64) ? 5 : (BM * BN <= 96 * 128) ? 3 : 2) igemm_conv_kernel(const IgemmArgs p) {
    static_assert(BM % 32 == 0 && BN % 32 == 0, "block tile must be a multiple of 32x32");
    static_assert(!XBN || (VEC && !BATCHED), "the BatchNorm-on-load form exists for the vector loader of a single convolution");
    constexpr int TM = BM / 32, TN = BN / 32;          // 16x16 MFMA tiles per wave (2x2 waves)
    constexpr int A_FLOATS = BM * LDK, B_FLOATS = BN * LDK;
    constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    // one LDS object: [2 stages of A|B] [row info int4 x BM].  The vector path only needs the row info in its prologue
    // (it ends up in registers as offsets + tap masks), so there it aliases stage 1, which is first written behind the
    // second barrier -- 96x64 then fits 5 blocks per CU (30.7 KB) instead of 4.
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_FLOATS + (VEC ? 0 : 4 * BM)];
    static_assert(!VEC || STAGE_FLOATS >= 4 * BM, "row info must fit in one stage");
    int4* rowinfo = reinterpret_cast<int4*>(smem + (VEC ? STAGE_FLOATS : 2 * STAGE_FLOATS));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile = blockIdx.x, piece = -1, kbeg = 0;
    int plane = blockIdx.y;
    if constexpr (BATCHED) {
        // Whole planes per XCD (round 4).  Workgroups are dispatched in linear order (x fastest) round-robin over the 8 XCDs, each with its
        // own L2; with blockIdx.y = plane every XCD touched EVERY plane's weight matrix U[g] (1 MB at 512 x 512) and the L2-miss traffic of a
        // launch was 232 MB read for ~120 MB of operands.  Here the j-th block an XCD receives works on plane k + 8 (j / tiles) -- a plane's
        // V and U live in ONE L2 --, and the planes left over when their count is not a multiple of 8 (36 = 4 x 8 + 4) are shared by
        // 8 / rem XCDs each, a contiguous range of row blocks per XCD.
        if (p.plane_xcd) {
            const unsigned gx = gridDim.x, P = gridDim.y;
            const unsigned lin = blockIdx.y * gx + blockIdx.x, k = lin & 7u, j = lin >> 3;
            const unsigned full = P >> 3, rem = P & 7u;
            if (j < full * gx) {
                plane = (int)(k + 8u * (j / gx));
                tile = (int)(j % gx);
            } else {
                const unsigned share = 8u / rem, jj = j - full * gx;      // jj < gx / share
                plane = (int)(8u * full + (k % rem));
                tile = (int)((k / rem) * (gx / share) + jj);
            }
        } else if (tile < 8 * p.xcd_chunk) {
            tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
        }
    } else {
        if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    }
    int nk = VEC ? (p.K / BK) : ((p.K + BK - 1) / BK);
    if constexpr (VEC && !BATCHED) {
        if (tile >= p.full_tiles) {                   // a K-slice of one of the last round's tiles
            piece = tile - p.full_tiles;
            tile = p.full_tiles + piece / p.splits;
            kbeg = (piece % p.splits) * p.kper;
            nk = min(nk - kbeg, p.kper);
        }
    }
    const int groupsN = BATCHED ? p.tilesN / p.nt_run : p.tilesN;
    const int tile_m = tile / groupsN, tile_n = (tile % groupsN) * (BATCHED ? p.nt_run : 1);
    const int m0 = tile_m * BM;
    int n0 = tile_n * BN;
    const int nk1 = nk;                               // K-steps of one tile
    if constexpr (BATCHED) nk *= p.nt_run;            // ... of the block's whole run of N-tiles

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

    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // ---- staging registers: TWO sets, so global loads run two K-steps ahead of the MFMAs ----
    constexpr int A_PASSES_V = (BM + 63) / 64, B_PASSES_V = (BN + 63) / 64;
    constexpr int A_PASSES_S = BM / 16, B_PASSES_S = BN / 16;
    float4 va[2][VEC ? A_PASSES_V : 1], vb[2][VEC ? B_PASSES_V : 1];
    float sa[2][VEC ? 1 : A_PASSES_S], sb[2][VEC ? 1 : B_PASSES_S];
    // XBN: scale | shift of every input channel in LDS (formed from the producing convolution's statistics, or the saved pair); per
    // register set the channel of the lane's first value and which passes loaded a real pixel -- out-of-image taps / rows past M must
    // stay 0 behind the BatchNorm + ReLU
    __shared__ __attribute__((aligned(16))) float xtab[XBN ? 2 * kCoeffMaxC : 4];
    int xch[2] = {0, 0};
    unsigned xok[2] = {0u, 0u};
    if constexpr (XBN) {
        if (p.in.live) {
            bn_coeff_block(p.in.bn, xtab);
        } else {
            for (int c = tid; c < p.Cin; c += 256) { xtab[c] = p.in.scale[c]; xtab[kCoeffMaxC + c] = p.in.shift[c]; }
        }
        __syncthreads();
    }

    const int lrow = tid >> 2, kq = tid & 3;    // VEC: 4 threads x float4 per row, 64 rows per pass
    const int srow = tid >> 4, kl = tid & 15;   // SCALAR: 16 threads per row, 16 rows per pass
    // VEC K order: channel chunk outer, filter tap inner -- the R*S taps of one 16-channel chunk re-read the
    // same 64-B lines of x (shifted rows), so they hit L1/L2 instead of streaming x once per tap.
    int tap = 0, c0 = 0, ky = 0, kx = 0;
    if constexpr (VEC) {
        if (kbeg) { tap = kbeg % (p.R * p.S); c0 = (kbeg / (p.R * p.S)) * BK; ky = tap / p.S; kx = tap - ky * p.S; }
    }
    // VEC: this thread's rows (pixel base / top-left input coordinate) live in registers; invalid rows can never
    // pass the bounds test
    const float* px = BATCHED ? p.x + plane * p.gx : p.x;
    const float* pw = BATCHED ? p.w + plane * p.gw : p.w;
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pw), 0, p.w_bytes, 0x00020000);
    // VEC: per row of this thread, the byte offset of tap (0,0) / channel 0 and the bit mask of the filter taps that fall
    // inside the image (R*S <= 32; rows past M have an empty mask) -- a K-step then costs one add, one mask test and one
    // select per load instead of re-deriving the input coordinates
    unsigned abase[VEC ? A_PASSES_V : 1], amask[VEC ? A_PASSES_V : 1], bbase[VEC ? B_PASSES_V : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES_V; ++ps) {
            const int r = lrow + 64 * ps;
            int4 ri = (BM % 64 == 0 || r < BM) ? rowinfo[r] : make_int4(0, 0, 0, 0);
            abase[ps] = (unsigned)(((ri.x + ri.y * p.Win + ri.z) * p.Cin + kq * 4) * 4);
            unsigned mk = 0u;
            if (ri.w)
                for (int t = 0; t < p.R * p.S; ++t) {
                    const int ty = t / p.S, tx = t - ty * p.S;
                    if ((unsigned)(ri.y + ty) < (unsigned)p.Hin && (unsigned)(ri.z + tx) < (unsigned)p.Win) mk |= 1u << t;
                }
            amask[ps] = mk;
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES_V; ++ps) {
            const int n = lrow + 64 * ps;
            bbase[ps] = (BN % 64 == 0 || n < BN) ? (unsigned)(((n0 + n) * p.K + kq * 4) * 4) : 0xFFFFFFFFu;
        }
    }

    auto load_tile = [&](auto set_c, int kt, bool live) {   // live == false: past the last K-tile, every lane reads zeros
        constexpr int SET = decltype(set_c)::value;
        if constexpr (VEC) {
            // branch-free: out-of-image taps / ragged rows get byte offset 0xFFFFFFFF, which the buffer bounds
            // check turns into a zero fill
            const unsigned soff_a = (unsigned)(((ky * p.Win + kx) * p.Cin + c0) * 4), soff_b = (unsigned)((tap * p.Cin + c0) * 4);
            const unsigned bit = live ? (1u << tap) : 0u;
            if constexpr (XBN) {
                xch[SET] = (c0 < p.Cin ? c0 : 0) + kq * 4;              // (past the last K-tile every pass is masked anyway)
                xok[SET] = 0u;
            }
#pragma unroll
            for (int ps = 0; ps < A_PASSES_V; ++ps) {
                const unsigned voff = (amask[ps] & bit) ? abase[ps] + soff_a : 0xFFFFFFFFu;
                va[SET][ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, 0));
                if constexpr (XBN) xok[SET] |= (amask[ps] & bit) ? (1u << ps) : 0u;
            }
#pragma unroll
            for (int ps = 0; ps < B_PASSES_V; ++ps) {
                const unsigned voff = (live && bbase[ps] != 0xFFFFFFFFu) ? bbase[ps] + soff_b : 0xFFFFFFFFu;
                vb[SET][ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, 0, 0));
            }
        } else {
            int k = kt * BK + kl;
            bool kok = live && k < p.K;
            int t = k / p.Cin, ci = k - t * p.Cin;
            int yy = t / p.S, xx = t - yy * p.S;
#pragma unroll
            for (int ps = 0; ps < A_PASSES_S; ++ps) {
                int4 ri = rowinfo[srow + 16 * ps];
                int iy = ri.y + yy, ix = ri.z + xx;
                bool ok = kok && ri.w && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                sa[SET][ps] = ok ? px[(size_t)(ri.x + iy * p.Win + ix) * p.Cin + ci] : 0.f;
            }
#pragma unroll
            for (int ps = 0; ps < B_PASSES_S; ++ps) {
                int n = srow + 16 * ps;
                sb[SET][ps] = kok ? pw[(size_t)(n0 + n) * p.K + k] : 0.f;
            }
        }
    };
    auto store_tile = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        float* As = smem + buf * STAGE_FLOATS;
        float* Bs = As + A_FLOATS;
        if constexpr (VEC) {
#pragma unroll
            for (int ps = 0; ps < A_PASSES_V; ++ps) {
                const int r = lrow + 64 * ps;
                float4 v = va[SET][ps];
                if constexpr (XBN) {
                    const bool ok = (xok[SET] >> ps) & 1u;
                    const float4 sc = *reinterpret_cast<const float4*>(xtab + xch[SET]), sh = *reinterpret_cast<const float4*>(xtab + kCoeffMaxC + xch[SET]);
                    v.x = ok ? fmaxf(__builtin_fmaf(v.x, sc.x, sh.x), 0.f) : 0.f; v.y = ok ? fmaxf(__builtin_fmaf(v.y, sc.y, sh.y), 0.f) : 0.f;
                    v.z = ok ? fmaxf(__builtin_fmaf(v.z, sc.z, sh.z), 0.f) : 0.f; v.w = ok ? fmaxf(__builtin_fmaf(v.w, sc.w, sh.w), 0.f) : 0.f;
                }
                if (BM % 64 == 0 || r < BM) *reinterpret_cast<float4*>(As + r * LDK + ((kq ^ lds_sw(r)) << 2)) = v;
            }
#pragma unroll
            for (int ps = 0; ps < B_PASSES_V; ++ps) {
                const int n = lrow + 64 * ps;
                if (BN % 64 == 0 || n < BN) *reinterpret_cast<float4*>(Bs + n * LDK + ((kq ^ lds_sw(n)) << 2)) = vb[SET][ps];
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASSES_S; ++ps) As[(srow + 16 * ps) * LDK + ((((kl >> 2) ^ lds_sw(srow + 16 * ps)) << 2) | (kl & 3))] = sa[SET][ps];
#pragma unroll
            for (int ps = 0; ps < B_PASSES_S; ++ps) Bs[(srow + 16 * ps) * LDK + ((((kl >> 2) ^ lds_sw(srow + 16 * ps)) << 2) | (kl & 3))] = sb[SET][ps];
        }
    };
    auto advance = [&]() {   // VEC: next K-tile position (tap inner, channel chunk outer)
        ++tap;
        ++kx;
        if (kx >= p.S) { kx = 0; ++ky; }
        if (tap >= p.R * p.S) { tap = 0; kx = 0; ky = 0; c0 += BK; }
        if constexpr (BATCHED) {
            if (c0 >= p.K) {                          // next N-tile of the run: same rows, the following BN weight rows
                c0 = 0;
#pragma unroll
                for (int ps = 0; ps < B_PASSES_V; ++ps)
                    if (bbase[ps] != 0xFFFFFFFFu) bbase[ps] += (unsigned)(BN * p.K * 4);
            }
        }
    };

    // v_mfma_f32_16x16x4_f32 operands: lane l holds A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]
    const int fi = lane & 15, fq = lane >> 4;
    // Software pipeline: while the MFMAs of K-step k run from register fragment set (k & 1), the same wave already
    // reads the fragments of step k+1 from LDS into the other set, has the global loads of tile k+3 in flight and
    // writes tile k+2 to the LDS stage that step k's fragments came from -- a wave never leaves the matrix pipe idle
    // waiting for LDS / HBM, so two waves per SIMD are enough to keep it saturated.
    floatx4 af[2][TM], bf[2][TN];
    auto read_frags = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        const float* As = smem + buf * STAGE_FLOATS;
        const float* Bs = As + A_FLOATS;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[SET][i] = *reinterpret_cast<const floatx4*>(As + (wm * (BM / 2) + i * 16 + fi) * LDK + ((fq ^ lds_sw(wm * (BM / 2) + i * 16 + fi)) << 2));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bf[SET][j] = *reinterpret_cast<const floatx4*>(Bs + (wn * (BN / 2) + j * 16 + fi) * LDK + ((fq ^ lds_sw(wn * (BN / 2) + j * 16 + fi)) << 2));
    };
    auto mfma_step = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e)   // MFMA e contracts k = {e, 4+e, 8+e, 12+e} (same permutation for A and B)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[SET][i][e], bf[SET][j][e], acc[i][j], 0, 0, 0);
    };
    // batched launches: plain bounds-checked buffer stores of the finished N-tile (rows past M get offset 0xFFFFFFFF and are dropped)
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(BATCHED ? p.epi.y + plane * p.gy : p.epi.y, 0,
                                                                     BATCHED ? (unsigned)p.M * (unsigned)p.Cout * 4u : 0u, 0x00020000);
    const int st_row0 = m0 + wm * (BM / 2) + 4 * fq;             // this lane's first output row; + i * 16 + r
    auto store_plain = [&](int ncol0) {
        const unsigned base = (unsigned)((st_row0 * p.Cout + ncol0 + wn * (BN / 2) + fi) * 4);
        const int left = p.M - st_row0;                           // rows of this lane's column that exist
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned rowoff = (i * 16 + r) < left ? base + (unsigned)((i * 16 + r) * p.Cout * 4) : 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float v = acc[i][j][r];     // (bit_cast straight from the vector element stored element 0 four times)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc, rowoff == 0xFFFFFFFFu ? rowoff : rowoff + j * 64, 0, 0);
                }
            }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // prologue.  All loads / stores are issued UNCONDITIONALLY (tiles past the end read zeros) so that the compiler
    // can count outstanding loads and wait only for the older register set at each LDS store.
    load_tile(S0{}, 0, true);                         // tile 0
    if constexpr (VEC) advance();
    load_tile(S1{}, 1, nk > 1);                       // tile 1
    store_tile(S0{}, 0);                              // tile 0 -> stage 0
    if constexpr (VEC) advance();
    load_tile(S0{}, 2, nk > 2);                       // tile 2 in flight (set 0)
    __syncthreads();
    read_frags(S0{}, 0);                              // fragments of step 0
    store_tile(S1{}, 1);                              // tile 1 -> stage 1
    if constexpr (VEC) advance();
    load_tile(S1{}, 3, nk > 3);                       // tile 3 in flight (set 1)
    __syncthreads();
    // invariant at the top of step k (k even shown): fragments(k) in set 0; stage 1 holds tile k+1 (visible);
    // stage 0 is free (every wave read fragments(k) before the last barrier); tile k+2 in global set 0, k+3 in set 1
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        read_frags(S1{}, 1);                          // fragments(k+1)
        mfma_step(S0{});                              // step k
        store_tile(S0{}, 0);                          // tile k+2 -> stage 0
        if constexpr (VEC) advance();
        load_tile(S0{}, kt + 4, kt + 4 < nk);
        __syncthreads();
        read_frags(S0{}, 0);                          // fragments(k+2)
        mfma_step(S1{});                              // step k+1
        store_tile(S1{}, 1);                          // tile k+3 -> stage 1
        if constexpr (VEC) advance();
        load_tile(S1{}, kt + 5, kt + 5 < nk);
        __syncthreads();
        if constexpr (BATCHED) {
            if ((kt + 2) % nk1 == 0 && kt + 2 < nk) { // an N-tile of the run is complete: store it, keep the pipeline going
                store_plain(n0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
                n0 += BN;
            }
        }
    }
    if (kt < nk) mfma_step(S0{});                     // odd tile count: last step's fragments are in set 0
    __syncthreads();

    // ---- epilogue (igemm_epilogue.h); the stage buffers are idle now and serve as its reduction scratch ----
    // (the LDS-staged float4 form of igemm_epilogue.h measured 1 % slower on the whole fp32 step: other blocks of the CU
    // cover a scalar epilogue with their MFMAs, and the strip round trip adds LDS traffic)
    if constexpr (VEC && !BATCHED) {
        if (piece >= 0) {                             // block-uniform: the fix-up kernel owns this tile's epilogue
            float* dst = p.partial + (size_t)piece * (BM * BN) + tid;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[((i * TN + j) * 4 + r) * 256] = acc[i][j][r];
            return;
        }
    }
    if constexpr (BATCHED) {                          // batched launch: plain store into the g-th output
        store_plain(n0);
        return;
    }
    igemm_epilogue<BM, BN, TM, TN>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
    if constexpr (XBN) inbn_commit(p.in);             // (block 0: mean / invstd / scale / shift for backward, running statistics)
}

// Sums the K-slices of the last round's tiles (fixed order: deterministic) and applies the epilogue the main kernel skipped.
template <int BM, int BN>
__global__ void __launch_bounds__(256) igemm_tail_fixup_kernel(const IgemmArgs p) {
    constexpr int TM = BM / 32, TN = BN / 32;
    __shared__ double red[2 * BN * 4];
    const int tid = threadIdx.x;
    const int tile = p.full_tiles + blockIdx.x;
    const int m0 = (tile / p.tilesN) * BM, n0 = (tile % p.tilesN) * BN;
    floatx4 acc[TM][TN];
    const float* src = p.partial + (size_t)blockIdx.x * p.splits * (BM * BN) + tid;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = src[((i * TN + j) * 4 + r) * 256];
    for (int c = 1; c < p.splits; ++c) {
        src += BM * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * TN + j) * 4 + r) * 256];
    }
    igemm_epilogue<BM, BN, TM, TN>(p.epi, acc, m0, n0, p.M, p.Cout, red);
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

// ---- balanced last round (opt-in: LaunchTune::tail_split; SIMQ_TAIL_SPLIT=1 in the ablation build) ----------------------------------------
// A launch of T tiles on S = 256 CUs x (resident blocks per CU) slots runs floor(T / S) full rounds and then a partial one
// in which T mod S blocks have the chip to themselves: a 29-sample next-state batch through a 512-channel layer is 1392
// tiles of 96x64 on 1280 slots, i.e. 112 CUs end with ONE block (4 waves).  With this switch the tiles of that last round
// are cut along K into `splits` slices (splits x (T mod S) <= S), which fill the slots the finished blocks of the last
// full round free up, with 1/splits of a tile each.  Slices leave raw accumulators in a per-stream scratch (plain stores
// in the kernel's own register order, fully coalesced); the fix-up kernel adds them in a fixed order (deterministic).
// Measured (tools/tail_split.py): isolated launches gain +9..13 % at B = 29 and +3 % at B = 32 on the 512-channel layers;
// the whole step LOSES 0.4 % (1782 vs 1790 tr/s, three alternating runs): the next-state forwards of the policy and the
// target net already run concurrently on two streams and fill each other's last rounds, and at B = 32 the gain does not
// pay for the extra launch.  Hence off by default; it is the better setting when the forwards run serialised.
constexpr int kNumCU = 256;
constexpr int kMaxTailSplits = 8;
constexpr size_t kTailScratchBytes = 48u << 20;   // >= S x BM x BN x 4 for every menu tile (1536 slots x 64x64 = 25 MB, 768 x 96x128 = 38 MB)

float* tail_scratch(hipStream_t stream) {
    struct Entry { int dev; hipStream_t stream; float* ptr; };
    static std::mutex mu;
    static std::vector<Entry> entries;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    for (const Entry& e : entries)
        if (e.dev == dev && e.stream == stream) return e.ptr;
    void* ptr = nullptr;
    if (hipMalloc(&ptr, kTailScratchBytes) != hipSuccess) { (void)hipGetLastError(); ptr = nullptr; }
    entries.push_back({dev, stream, static_cast<float*>(ptr)});
    return static_cast<float*>(ptr);
}

int g_xcd_remap = -1;    // SIMQ_XCD_REMAP=0: tiles in launch order (A-B runs, ablation build)

template <int BM, int BN, bool VEC, bool BATCHED = false>
int resident_blocks() {
    static int cached = 0;
    if (!cached) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, igemm_conv_kernel<BM, BN, VEC, BATCHED>, 256, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        cached = n;
    }
    return cached;
}

template <int BM, int BN, bool VEC, bool BATCHED = false>
int run(const IgemmArgs& a, hipStream_t stream, const LaunchTune& tune, int batch = 1) {
    if constexpr (!VEC || BATCHED) {
        SIMQ_REQUIRE(!a.in.on(), "conv_igemm: BatchNorm-on-load needs the vector loader (Cin %% 16 == 0) of a single convolution");
    }
    IgemmArgs p = a;
    p.tilesN = p.Cout / BN;
    int tilesM = (p.M + BM - 1) / BM;
    const int tiles = tilesM * p.tilesN;
    p.full_tiles = tiles; p.splits = 1; p.kper = 0; p.partial = nullptr;
    int tail = 0;
    if constexpr (VEC && !BATCHED) {
        static const int env_tail_split = SIMQ_TUNE_INT("SIMQ_TAIL_SPLIT", 0) != 0 ? 1 : 0;      // (ablation build only)
        const int tail_split = tune.tail_split != 0 ? 1 : env_tail_split;
        const int slots = kNumCU * resident_blocks<BM, BN, VEC>();
        const int rem = tiles % slots, nk = p.K / BK;
        // worth it when the last round leaves a CU with one or two blocks (three or more co-resident blocks already keep the
        // matrix pipe busy: slicing a half-full round of the 64x64 tile measured 4 % slower) and a slice still has a
        // pipeline's worth of K-steps
        if (tail_split && tiles > slots && rem > 0 && rem * 2 <= kNumCU * 3) {
            int s = slots / rem;
            if (s > kMaxTailSplits) s = kMaxTailSplits;
            while (s > 1 && nk / s < 24) --s;
            if (s > 1 && (size_t)rem * s * BM * BN * 4 <= kTailScratchBytes && (p.partial = tail_scratch(stream)) != nullptr) {
                p.full_tiles = tiles - rem;
                p.splits = s;
                p.kper = (nk + s - 1) / s;
                tail = rem;
            }
        }
    }
    p.nt_run = 1;
    int launch_tiles = p.full_tiles;
    if constexpr (BATCHED) {
        // N-tiles per block (SIMQ_GEMM_NT_RUN, default 1).  Measured on the headline step: runs of 1 / 2 / 4 N-tiles give
        // 2422 / 2426 / 2391 tr/s -- carrying the pipeline across tiles does NOT recover the short-K loss (the shortk probe's
        // 97 -> 130 TF/s from K = 256 to 1024 comes with 4x fewer output bytes per flop, not only fewer prologues), and
        // longer runs cost block-level parallelism.  Kept as a switch for other shapes.
        static const int forced = SIMQ_TUNE_INT("SIMQ_GEMM_NT_RUN", 1);
        if ((p.K / BK) % 2 == 0 && forced >= 1 && p.tilesN % forced == 0) p.nt_run = forced;
        launch_tiles = tilesM * (p.tilesN / p.nt_run);
        p.full_tiles = launch_tiles;
    }
    if (g_xcd_remap < 0) g_xcd_remap = SIMQ_TUNE_INT("SIMQ_XCD_REMAP", 1) != 0 ? 1 : 0;
    p.xcd_chunk = (g_xcd_remap && p.tilesN / p.nt_run > 1 && p.full_tiles >= 64) ? p.full_tiles / 8 : 0;
    p.plane_xcd = 0;
    if constexpr (BATCHED) {
        const int rem = batch & 7;
        const bool rem_ok = rem == 0 || ((rem == 1 || rem == 2 || rem == 4) && p.full_tiles % (8 / rem) == 0);
        if (tune.plane_xcd && g_xcd_remap && batch >= 8 && rem_ok && ((long)p.full_tiles * batch) % 8 == 0) p.plane_xcd = 1;
    }
    dim3 grid((unsigned)(p.full_tiles + tail * p.splits), (unsigned)batch);
    // algorithmic work: 2*M*N*K flops; bytes = read x once + read w once + write y once
    // profiling kinds: 0 = the dominant kernel of the headline workload (the batched transform-domain GEMM of the Winograd
    // layers), 2 = every other implicit-GEMM launch, 1 = wgrad
    note_launch(BATCHED ? "gemm_f32_batched" : (VEC ? "igemm_f32" : "igemm_f32_gather"));
    prof_launch_begin(BATCHED ? 0 : 2, 2.0 * p.M * p.Cout * p.K * batch,
                      4.0 * batch * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    if constexpr (VEC && !BATCHED) {
        SIMQ_REQUIRE(!p.in.on() || p.Cin <= kCoeffMaxC, "conv_igemm: BatchNorm-on-load holds at most %d input channels (Cin=%d)", kCoeffMaxC, p.Cin);
        if (p.in.on()) hipLaunchKernelGGL((igemm_conv_kernel<BM, BN, VEC, BATCHED, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((igemm_conv_kernel<BM, BN, VEC, BATCHED>), grid, dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((igemm_conv_kernel<BM, BN, VEC, BATCHED>), grid, dim3(256), 0, stream, p);
    }
    if (tail) hipLaunchKernelGGL((igemm_tail_fixup_kernel<BM, BN>), dim3((unsigned)tail), dim3(256), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

// ---- tile selection -------------------------------------------------------------------------------------
// The 256 CUs share nothing, so the launch time is (max blocks on one CU) x (time of one block).  Pick the
// (BM, BN) of the menu that minimises  ceil(blocks / 256) * BM * BN / efficiency(BM, BN):  bigger tiles reuse
// operands better, but a block count that is not a multiple of 256 leaves CUs idle in the last round.
// relative per-tile efficiency, measured on the 512-channel 3x3 layer (tools/tune_conv.py).  `eff`: a few blocks per CU
// (B = 32: M = 18432) -- the 96x64 tile (5 resident blocks per CU since its row info left LDS; 131 vs 126 TF/s for 96x128) leads; `eff_many`: >= 8 blocks per CU (B >= 64), where the
// narrower tiles with 4-6 resident blocks per CU take over (128x64 reaches 140.7 TF/s = 0.89 of the fp32 matrix peak).  The
// 128x128 tile needs 182 registers (2 waves per SIMD) and trails in both regimes.
struct TileCfg { int bm, bn; float eff, eff_many; };
constexpr TileCfg kMenu[] = {
    {96, 128, 1.00f, 0.934f}, {96, 64, 1.05f, 0.984f}, {64, 64, 0.95f, 0.974f}, {64, 128, 0.92f, 0.959f}, {128, 64, 0.88f, 1.00f},
    {128, 32, 0.86f, 0.915f}, {64, 32, 0.86f, 0.877f},  {32, 64, 0.86f, 0.854f}, {96, 32, 0.84f, 0.882f}, {128, 128, 0.75f, 0.70f},
    {32, 32, 0.72f, 0.75f},
};
// a forced tile: LaunchTune::force_bm / force_bn of the launch (simq_launch_opts of a standalone operator call: per-kernel tests,
// tools/tune_conv.py), or SIMQ_IGEMM_TILE=BMxBN in the ablation build
int forced_tile(const LaunchTune& t, int* bm, int* bn) {
    if (t.force_bm > 0) { *bm = t.force_bm; *bn = t.force_bn; return 1; }
#ifdef SIMQ_ABLATIONS
    static int env_bm = -1, env_bn = 0;
    if (env_bm == -1) {
        env_bm = 0;
        const char* s = getenv("SIMQ_IGEMM_TILE");
        if (s && sscanf(s, "%dx%d", &env_bm, &env_bn) != 2) env_bm = 0;
    }
    if (env_bm > 0) { *bm = env_bm; *bn = env_bn; return 1; }
#endif
    *bm = 0; *bn = 0;
    return 0;
}

template <bool VEC>
int dispatch(int bm, int bn, const IgemmArgs& a, hipStream_t stream, const LaunchTune& tune) {
#define SIMQ_TILE(BM_, BN_) if (bm == BM_ && bn == BN_) return run<BM_, BN_, VEC>(a, stream, tune)
    if constexpr (VEC) {
        SIMQ_TILE(128, 128); SIMQ_TILE(96, 128); SIMQ_TILE(64, 128); SIMQ_TILE(128, 64); SIMQ_TILE(96, 64);
        SIMQ_TILE(64, 64); SIMQ_TILE(128, 32); SIMQ_TILE(96, 32); SIMQ_TILE(64, 32); SIMQ_TILE(32, 64); SIMQ_TILE(32, 32);
    } else {
        SIMQ_TILE(128, 64); SIMQ_TILE(64, 64); SIMQ_TILE(32, 64);
    }
#undef SIMQ_TILE
    set_error("conv_igemm: no kernel for tile %dx%d", bm, bn);
    return -1;
}

}  // namespace

int launch_conv_igemm(const float* x, const float* w, float* y, const ConvGeom& g, const ConvEpilogue& e,
                      hipStream_t stream, const InBn& in) {
    IgemmArgs a;
    a.x = x; a.w = w;
    a.in = in;
    a.epi = make_epi(y, e);
    a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.R = g.R; a.S = g.S; a.stride = g.stride; a.pad = g.pad;
    a.M = g.M(); a.K = g.K(); a.tilesN = 0;
    a.gx = a.gw = a.gy = 0; a.xcd_chunk = 0; a.nt_run = 1; a.plane_xcd = 0;
    SIMQ_REQUIRE(a.M > 0, "conv: empty problem");
    const double xb = 4.0 * g.B * g.Hin * g.Win * g.Cin, wb = 4.0 * g.Cout * a.K;
    SIMQ_REQUIRE(xb < 4294967000.0 && wb < 4294967000.0, "conv_igemm: tensor exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    SIMQ_REQUIRE(g.Cout % 32 == 0, "conv_igemm: Cout=%d must be a multiple of 32", g.Cout);
    const bool vec = (g.Cin % BK) == 0 && g.R * g.S <= 32;   // vector path: 16-channel chunks, tap validity kept as a 32-bit mask
    SIMQ_REQUIRE(vec || g.Cout % 64 == 0, "conv_igemm (generic gather): Cout=%d must be a multiple of 64", g.Cout);
    int bm = 0, bn = 0;
    const bool forced = forced_tile(g.tune, &bm, &bn) != 0;
    if (!forced) {                       // the 64-input-channel 3x3 layers on the 24x24 maps: image-tile kernel (conv_img_f32.hip)
        if (int rc = try_conv_img_f32(x, w, y, g, e, stream, in)) return rc < 0 ? rc : 0;
    }
    if (!forced || g.Cout % bn != 0 || (!vec && !(bn == 64 && (bm == 128 || bm == 64 || bm == 32)))) {
        double best = 1e300;
        for (const TileCfg& t : kMenu) {
            if (g.Cout % t.bn != 0) continue;
            if (!vec && !(t.bn == 64 && (t.bm == 128 || t.bm == 64 || t.bm == 32))) continue;
            const long blocks = (long)((a.M + t.bm - 1) / t.bm) * (g.Cout / t.bn);
            const long rounds = (blocks + kNumCU - 1) / kNumCU;
            double cost = (double)rounds * t.bm * t.bn / (rounds >= 8 ? t.eff_many : t.eff);
            if (rounds == 1) cost *= 1.25;      // a lone block per CU cannot hide its barriers behind another block
            if (cost < best) { best = cost; bm = t.bm; bn = t.bn; }
        }
    }
    return vec ? dispatch<true>(bm, bn, a, stream, g.tune) : dispatch<false>(bm, bn, a, stream, g.tune);
}

// `batch` independent GEMMs  y_g[M][N] = x_g[M][K] * w_g[N][K]^T  (row-major, g-th operand at base + g * rows * cols) in one
// launch (grid.y = batch): the transform-domain contractions of conv_winograd.hip.  K % 16 == 0, N % 64 == 0.
int launch_gemm_batched(const float* x, const float* w, float* y, int M, int N, int K, int batch, hipStream_t stream, const LaunchTune& tune) {
    SIMQ_REQUIRE(M > 0 && K % BK == 0 && N % 64 == 0 && batch >= 1, "gemm_batched: M=%d N=%d K=%d batch=%d not supported", M, N, K, batch);
    if (tune.gemm_split == 1 && gemm_split3_eligible(M, N, K, batch)) return launch_gemm_batched_split3(x, w, y, M, N, K, batch, stream, tune);
#ifdef SIMQ_ABLATIONS      // the opt-in ping-pong form (gemm_f32_pp.hip, step-neutral: docs/history.md 4) exists in libsimq_ablate.so only
    if (int rc = try_gemm_batched_pp(x, w, y, M, N, K, batch, stream)) return rc < 0 ? rc : 0;     // (N % 128 == 0)
#endif
    IgemmArgs a;
    a.x = x; a.w = w;
    ConvEpilogue e;
    a.epi = make_epi(y, e);
    a.Hin = M; a.Win = 1; a.Cin = K; a.Hout = M; a.Wout = 1; a.Cout = N; a.R = 1; a.S = 1; a.stride = 1; a.pad = 0;
    a.M = M; a.K = K; a.tilesN = 0;
    a.gx = (long)M * K; a.gw = (long)N * K; a.gy = (long)M * N; a.xcd_chunk = 0; a.nt_run = 1; a.plane_xcd = 0;
    const double xb = 4.0 * M * K, wb = 4.0 * N * K;
    SIMQ_REQUIRE(xb < 4294967000.0 && wb < 4294967000.0, "gemm_batched: operand exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    // many rounds of short-K blocks: the 64x64 tile (6 resident blocks per CU) measured best on these shapes
    // (tools/probes/wino_gemm_probe.py: 104-107 TF/s at K = 256..512), 96x64 when the rows do not fill 64-row tiles evenly
    // (64x32 / 32x32 tiles for the few-block transform-domain weight gradients -- 256 blocks of a 256 x 256 x 16 problem -- measured
    // slower than one 64x64 block per CU: 0.181 vs 0.150 ms)
#ifdef SIMQ_ABLATIONS      // tile A/B for the transform-domain GEMMs (tools/ab_step.py): SIMQ_GEMM_BATCHED_TILE=96 | 128 (rows; 64 columns).
    // Whole fp32 step, three alternating runs each: 64 rows 3320 tr/s, 96 rows 3267, 128 rows 3269 -- the 64x64 tile stays.
    static const int bt = SIMQ_TUNE_INT("SIMQ_GEMM_BATCHED_TILE", 64);
    if (bt == 96 && M % 96 == 0) return run<96, 64, true, true>(a, stream, tune, batch);
    if (bt == 128 && M % 128 == 0) return run<128, 64, true, true>(a, stream, tune, batch);
#endif
    return run<64, 64, true, true>(a, stream, tune, batch);
}

int tune_forced_tile(const LaunchTune& t, int* bm, int* bn) { return forced_tile(t, bm, bn); }

int launch_weight_transpose(const float* w, float* wt, int cout, int taps, int cin, hipStream_t stream) {
    size_t total = (size_t)cout * taps * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, stream, w, wt, cout, taps, cin);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
