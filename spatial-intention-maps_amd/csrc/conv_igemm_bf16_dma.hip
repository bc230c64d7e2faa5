// Large-tile implicit-GEMM convolution on the bf16 matrix cores of gfx950, staged through LDS by the LDS-DMA path
// (buffer_load_dwordx4 ... lds), for plain bf16 operands (NP = 1).  Same contraction, operand layout and epilogue as
// conv_igemm_bf16.hip (reference operators: the nn.Conv2d forwards of resnet.py:94-102 / networks.py:18-26 and
// their dgrads, train.py:132); what changes is how the block is fed:
//
//   * One block per CU (production: 512 threads = 8 wave64 as 2 x 4, wave tile 144 x 32) owns a BM x BN tile with BM a
//     multiple of 144: the 24x24 feature maps give M = B*576 = 4*B*144 rows, so 288x128 tiles cut a 512-channel layer
//     of a 32-transition minibatch into exactly 256 tiles -- one full round of the 256 CUs, where the 96/128-row
//     tiles of the register-staged kernel need 1.5 - 2.25 rounds.
//   * K-step 64 (one 128-B line per operand row and stage).  Every wave instruction of the stager moves 8 rows x
//     128 B from L2 straight into LDS; the per-lane SOURCE offset carries the XOR swizzle (slot ^ ((row >> 1) & 7)),
//     because the DMA writes lane-linearly (wave-uniform base + 16 * lane).  Out-of-image taps, ragged rows and
//     tiles past the end use offset 0xFFFFFFFF: the buffer range check makes the DMA write zeros (probed on
//     MI355X, tools/probes/glds_probe.hip).  No staging VGPRs, no ds_write.
//   * 3 LDS stages (2 when 3 do not fit 160 KB): the DMA of tile k+3 is issued right behind the mid-stage barrier of
//     tile k (into the stage that barrier retired) and is waited for with a counted vmcnt two stages later, before a
//     raw s_barrier (a __syncthreads() would drain the DMA queue).
//   * Fragments are double-buffered in registers; the ds_read_b128 of k-step t+1 and the DMA pieces are spread behind
//     the MFMA rows of k-step t (all waves of the block run in phase, so a wave must overlap its own staging).
//   * Epilogue through LDS strips with 16-byte global accesses (igemm_epilogue_staged).
// Measured (MI355X, layer4 3x3 512->512, B = 32, 87 GFLOP): 95 us = 0.92 PFLOP/s vs 132 us for the register-staged
// kernel; ablations (tools/bf16_tiles.py, SIMQ_BF16_DBG) put the MFMAs alone at ~36 us, the fragment reads at ~31 us
// and the epilogue at ~10 us -- heavy ds_read traffic and MFMA issue overlap only partly on this CU.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int BM, int BN, int NW>
struct DmaCfg {
    static constexpr int ROWB = 128;                               // bytes per LDS row (64 bf16)
    static constexpr int GROUPS = (BM + BN) / 8;                   // 8-row x 128-B pieces per stage
    static constexpr int NI = (GROUPS + NW - 1) / NW;              // DMA instructions per wave and stage
    static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    static constexpr bool RAGGED = (GROUPS % NW) != 0;              // some waves issue one dummy piece (into `scratch`)
    static constexpr int NBUF = (3 * STAGE_BYTES + (RAGGED ? 1024 : 0) <= 160 * 1024) ? 3 : 2;
    static constexpr int SMEM = NBUF * STAGE_BYTES + (RAGGED ? 1024 : 0);
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// WM x WN waves (WM * WN = NW); wave tile (BM / WM) x (BN / WN).
// AIP ("A in place"): the activation fragments are NOT double-buffered -- fragment i of the next k-step is loaded into its own
// registers right behind the MFMA row that consumed it (it is needed again a whole half stage later), only the weight
// fragments keep two sets.  That frees TM * 4 registers and is what lets the 288 x 256 tile (wave tile 144 x 64: 144 accumulator
// registers) fit the 256-register budget of two waves per SIMD.
template <int BM, int BN, int WM, int NW, int DBG = 0, bool AIP = false>
__global__ void __launch_bounds__(NW * 64, 1) igemm_bf16_dma_kernel(const IgemmBfArgs p) {
    using C = DmaCfg<BM, BN, NW>;
    constexpr int WN = NW / WM;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int NI = C::NI, NBUF = C::NBUF, STAGE = C::STAGE_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[C::SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int tile = blockIdx.x;
    if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    const int tile_m = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int taps = p.R * p.S;

    // ---- stager state: piece g = i * 4 + wave covers stage rows 8g .. 8g+7; this lane moves physical 16-B slot
    // (lane & 7) of row 8g + (lane >> 3), i.e. logical k-chunk (lane & 7) ^ ((row >> 1) & 7) of that row
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[0]), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[0]), 0, p.w_bytes, 0x00020000);
    unsigned vbase[NI], vmask[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = i * NW + wave;
        const int rs = g * 8 + (lane >> 3);
        vbase[i] = 0u;
        vmask[i] = 0u;
        if (g < BM / 8) {                                   // activation rows (wave-uniform branch)
            const int r = rs, m = m0 + r;
            const int kq = (lane & 7) ^ ((r >> 1) & 7);
            if (m < p.M) {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                vbase[i] = (unsigned)((((b * p.Hin + iy0) * p.Win + ix0) * p.Cin + kq * 8) * 2);
                unsigned mk = 0u;
                for (int t = 0; t < taps; ++t) {
                    const int ky = t / p.S, kx = t - ky * p.S;
                    if ((unsigned)(iy0 + ky) < (unsigned)p.Hin && (unsigned)(ix0 + kx) < (unsigned)p.Win) mk |= 1u << t;
                }
                vmask[i] = mk;
            }
        } else if (g < C::GROUPS) {                         // weight rows
            const int n = rs - BM;
            const int kq = (lane & 7) ^ ((n >> 1) & 7);
            vbase[i] = (unsigned)(((n0 + n) * p.K + kq * 8) * 2);
            vmask[i] = 0xFFFFFFFFu;
        }
    }
    int tap = 0, c0 = 0, ky = 0, kx = 0;    // K order: 64-channel chunk outer, filter tap inner
    auto issue_tile = [&](int buf, bool live) {
        const unsigned soff_a = (unsigned)(((ky * p.Win + kx) * p.Cin + c0) * 2);
        const unsigned soff_b = (unsigned)((tap * p.Cin + c0) * 2);
        const unsigned bit = live ? (1u << tap) : 0u;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = i * NW + wave;
            bool is_a;
            if constexpr ((BM / 8) % NW == 0) is_a = i < BM / (8 * NW);   // same split for all waves
            else is_a = g < BM / 8;                                    // wave-uniform
            // the tap / channel-chunk offset goes into the per-lane offset (the range check looks at it alone, and
            // vbase of a border pixel is "negative" until the tap offset is added)
            const unsigned voff = (vmask[i] & bit) ? vbase[i] + (is_a ? soff_a : soff_b) : 0xFFFFFFFFu;
            char* dst = (C::RAGGED && g >= C::GROUPS) ? smem + NBUF * STAGE : smem + buf * STAGE + g * 1024;
            if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 16, voff, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)dst, 16, voff, 0, 0, 0);
        }
        ++tap; ++kx;
        if (kx >= p.S) { kx = 0; ++ky; }
        if (tap >= taps) { tap = 0; kx = 0; ky = 0; c0 += 64; }
    };

    // ---- fragment addressing: v_mfma_f32_16x16x32_bf16 lane l holds A[i = l & 15][k = 8 * (l >> 4) .. +7] (B alike).
    // (row >> 1) & 7 of row = base16 + fi does not depend on the 16-row tile index, so two lane constants per operand
    // (k-step 0 / 1 of the stage) plus compile-time tile offsets address every fragment.
    const int fi = lane & 15, fq = lane >> 4;
    const int ra = wm * WTM + fi, rb = wn * WTN + fi;
    const int sa = (ra >> 1) & 7, sb = (rb >> 1) & 7;
    int a_off[2], b_off[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        a_off[sub] = ra * 128 + (((sub * 4 + fq) ^ sa) << 4);
        b_off[sub] = BM * 128 + rb * 128 + (((sub * 4 + fq) ^ sb) << 4);
    }
    bf16x8 af[AIP ? 1 : 2][TM], bf[2][TN];
    auto read_frags = [&](auto set_c, int buf, int sub) {
        constexpr int SET = decltype(set_c)::value;
        const char* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[AIP ? 0 : SET][i] = *reinterpret_cast<const bf16x8*>(st + a_off[sub] + i * 2048);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[SET][j] = *reinterpret_cast<const bf16x8*>(st + b_off[sub] + j * 2048);
    };
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    const int nk = (DBG & 32) ? 0 : p.K / 64;     // DBG bits: timing ablations (tools/bf16_tiles.py), never dispatched in production
    // One half stage: the TM x TN MFMAs of fragment set FS, with the `NITEMS` staging operations of the half (fragment
    // reads of the next k-step, and in the second half the DMA pieces of tile kt + NBUF) spread behind the MFMA rows,
    // so a wave never sits in a block of ds_read / buffer_load with an idle matrix pipe (all waves of the block are in
    // phase behind the stage barrier, they cannot cover for each other).
    unsigned soff_a = 0, soff_b = 0, tbit = 0;
    auto dma_item = [&](int i, int dbuf) {
        const int g = i * NW + wave;
        bool is_a;
        if constexpr ((BM / 8) % NW == 0) is_a = i < BM / (8 * NW);
        else is_a = g < BM / 8;
        const unsigned voff = (vmask[i] & tbit) ? vbase[i] + (is_a ? soff_a : soff_b) : 0xFFFFFFFFu;
        char* dst = (C::RAGGED && g >= C::GROUPS) ? smem + NBUF * STAGE : smem + dbuf * STAGE + g * 1024;
        if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 16, voff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)dst, 16, voff, 0, 0, 0);
    };
    auto half_stage = [&](auto fs_c, auto dma_c, int rbuf, int rsub, int dbuf) {
        constexpr int FS = decltype(fs_c)::value;
        constexpr bool DMA = decltype(dma_c)::value;
        constexpr int AS = AIP ? 0 : FS;                             // fragment set the MFMAs read their A operand from
        constexpr int NITEMS = (AIP ? 0 : TM) + TN + (DMA ? NI : 0);
        constexpr int SLOTS = TM - 1;                               // after MFMA rows 0 .. TM-2
        constexpr int PER = (NITEMS + SLOTS - 1) / SLOTS;
        const char* st = smem + rbuf * STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (DBG & 16) asm volatile("" :: "v"(af[AS][i]), "v"(bf[FS][j]));
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[AS][i], bf[FS][j], acc[i][j], 0, 0, 0);
            }
            if constexpr (AIP) {                                    // refill this row's A fragment for the next k-step
                if constexpr (!(DBG & 8)) af[0][i] = *reinterpret_cast<const bf16x8*>(st + a_off[rsub] + i * 2048);
            }
            if (i < SLOTS) {
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int it = i * PER + q;
                    if (it < TN) { if constexpr (!(DBG & 8)) bf[FS ^ 1][it] = *reinterpret_cast<const bf16x8*>(st + b_off[rsub] + it * 2048); }
                    else if (!AIP && it < TN + TM) { if constexpr (!(DBG & 8)) af[AIP ? 0 : (FS ^ 1)][it - TN] = *reinterpret_cast<const bf16x8*>(st + a_off[rsub] + (it - TN) * 2048); }
                    else if (it < NITEMS) { if constexpr (!(DBG & 1)) dma_item(it - TN - (AIP ? 0 : TM), dbuf); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using DmaOn = std::integral_constant<bool, true>;
    using DmaOff = std::integral_constant<bool, false>;

    // prologue: tiles 0 .. NBUF-1 in flight, tile 0 landed and visible, fragments of (tile 0, k-step 0) in set 0
#pragma unroll
    for (int t = 0; t < NBUF; ++t) issue_tile(t, t < nk);
    wait_vmcnt<(NBUF - 1) * NI>();
    __builtin_amdgcn_s_barrier();
    read_frags(S0{}, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nbuf = (buf + 1 == NBUF) ? 0 : buf + 1;
        // k-step 0 of tile kt; fragments of k-step 1 are read behind its MFMA rows
        half_stage(S0{}, DmaOff{}, buf, 1, 0);
        // tile kt + 1 landed (this wave's pieces; NBUF - 2 younger tiles stay in flight); all reads of stage `buf` done
        if constexpr (!(DBG & 4)) wait_vmcnt<(NBUF - 2) * NI>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        // k-step 1 of tile kt; fragments of (tile kt + 1, k-step 0) and the DMA of tile kt + NBUF into the stage just
        // retired are issued behind its MFMA rows
        soff_a = (unsigned)(((ky * p.Win + kx) * p.Cin + c0) * 2);
        soff_b = (unsigned)((tap * p.Cin + c0) * 2);
        tbit = (kt + NBUF < nk) ? (1u << tap) : 0u;
        half_stage(S1{}, DmaOn{}, nbuf, 0, buf);
        ++tap; ++kx;
        if (kx >= p.S) { kx = 0; ++ky; }
        if (tap >= taps) { tap = 0; kx = 0; ky = 0; c0 += 64; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        buf = nbuf;
    }
    wait_vmcnt<0>();
    __syncthreads();

    if constexpr (TM % 3 == 0 && C::SMEM >= staged_epilogue_smem<BN, TN, WM, NW, 3>()) igemm_epilogue_staged<BM, BN, TM, TN, WM, NW, 3, true>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
    else igemm_epilogue<BM, BN, TM, TN, WM, NW>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
}

template <int BM, int BN, int WM, int NW = 4, int DBG = 0, bool AIP = false>
int run(const IgemmBfArgs& a, hipStream_t stream) {
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = bf16_xcd_chunk(((p.M + BM - 1) / BM) * p.tilesN, p.tilesN);
    const int tilesM = (p.M + BM - 1) / BM;
    note_launch("igemm_bf16_dma");
    prof_launch_begin(2, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout),
                      stream);
    hipLaunchKernelGGL((igemm_bf16_dma_kernel<BM, BN, WM, NW, DBG, AIP>), dim3((unsigned)(tilesM * p.tilesN)), dim3(NW * 64), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

int dispatch(int bm, int bn, const IgemmBfArgs& a, hipStream_t stream) {
    if (bm == 288 && bn == 128) {
#ifdef SIMQ_ABLATIONS      // timing ablations (tools/bf16_tiles.py): compiled into libsimq_ablate.so only
        if (const int v = SIMQ_TUNE_INT("SIMQ_BF16_DBG", 0)) {
#define SIMQ_DBG(N) if (v == N) return run<288, 128, 2, 8, N>(a, stream)
            SIMQ_DBG(1); SIMQ_DBG(2); SIMQ_DBG(8); SIMQ_DBG(16); SIMQ_DBG(23); SIMQ_DBG(32);
            if (v == 100) return run<288, 128, 2, 4, 0>(a, stream);      // four-wave variant (wave tile 144 x 64)
#undef SIMQ_DBG
        }
#endif
        return run<288, 128, 2, 8>(a, stream);
    }
    if (bm == 288 && bn == 256) return run<288, 256, 2, 8, 0, true>(a, stream);
    if (bm == 144 && bn == 128) return run<144, 128, 1>(a, stream);
    if (bm == 288 && bn == 64) return run<288, 64, 2>(a, stream);
    if (bm == 144 && bn == 64) return run<144, 64, 1>(a, stream);
    return 0;
}

}  // namespace

int try_conv_igemm_bf16_dma(const IgemmBfArgs& a, hipStream_t stream) {
    if (a.Cin % 64 != 0 || a.Cout % 64 != 0 || a.R * a.S > 32) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_BF16_DMA", 1);   // 0 = off
    if (mode == 0) return 0;
    int fbm = 0, fbn = 0;
    if (bf16_forced_tile(a, &fbm, &fbn)) return (a.Cout % fbn == 0) ? dispatch(fbm, fbn, a, stream) : 0;
    // Measured on MI355X (tools/bf16_tiles.py, B = 32 / 29): the 288x128 eight-wave tile wins where it fills whole rounds
    // of the 256 CUs -- the 512-channel layers (95 vs 132 us on layer4's 3x3, 92 vs 130 us at 29 samples); on the
    // 256/128-channel layers the register-staged 96-row tiles (3-4 blocks per CU) stay ahead, so only that case is taken.
    int bm = 0, bn = 0;
    static const int min_k = SIMQ_TUNE_INT("SIMQ_BF16_DMA_MINK", 64);
    if (a.Cout % 128 == 0 && a.K >= min_k) {
        const long blocks = (long)((a.M + 287) / 288) * (a.Cout / 128);
        const long rounds = (blocks + 255) / 256;
        if (blocks >= 200 && (double)blocks / (double)(rounds * 256) >= 0.85) { bm = 288; bn = 128; }
    }
    // 64 output channels (layer1 at large batches): the 144 x 64 four-wave tile, two blocks per CU (512 blocks at B = 128).  These
    // launches are bound by their epilogue traffic and by the operand volume staged per flop; against the register-staged 96 x 64 tile
    // +1.0 % on the bf16 configs[2] step, +1.4 % forward + backward (the 288 x 64 tile: neutral).  SIMQ_BF16_DMA_N64=0: off.
    static const int n64 = SIMQ_TUNE_INT("SIMQ_BF16_DMA_N64", 1);
    if (!bm && n64 && a.Cout == 64 && a.K >= min_k && a.M >= 144 * 400) { bm = n64 == 2 ? 288 : 144; bn = 64; }
    return bm ? dispatch(bm, bn, a, stream) : 0;
}

}  // namespace simq
