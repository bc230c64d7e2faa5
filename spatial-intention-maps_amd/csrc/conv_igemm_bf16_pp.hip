// 288 x 256 "ping-pong" implicit-GEMM convolution on the bf16 matrix cores of gfx950 (plain bf16 operands, NP = 1).
// Same contraction, operand layout, LDS-DMA staging idea and epilogue as conv_igemm_bf16_dma.hip (reference operators: the
// nn.Conv2d forwards of resnet.py:94-102 / networks.py:18-26 and their dgrads, train.py:132); what changes is WHEN each
// wave does what:
//
//   * conv_igemm_bf16_dma.hip runs its 8 waves in phase: every wave interleaves fragment reads, DMA issue and MFMAs, and the
//     ablations of that kernel (profiles/r01_bf16_dma_ablation_l4_b32.txt) show the three hardly overlapping (MFMAs alone
//     36 us, fragment reads alone 31 us, staging alone ~25 us, all together 85 us on layer4 at B = 32).
//   * Here the two waves that share a SIMD (wave w and w + 4: wave groups 0 / 1 = upper / lower half of the tile rows) run
//     ONE BARRIER APART.  Every K-tile (32 deep = one MFMA k-step) is a load segment (13 x ds_read_b128: the wave's 9 activation
//     and 4 weight fragments; this wave's 4-5 LDS-DMA pieces of the K-tile three ahead; counted vmcnt; lgkmcnt(0)) | s_barrier |
//     an MFMA segment (36 x v_mfma_f32_16x16x32_bf16 under s_setprio 1, no memory instruction) | s_barrier.  Because group 1
//     enters the loop one barrier late, a SIMD always has one wave in its MFMA segment while the other reads / stages: the
//     matrix pipe never waits for LDS or for DMA issue, and fragments need no double buffering (accumulators 9 x 4 tiles =
//     144 registers + 13 fragments = 52).
//   * Tile 288 x 256 (wave tile 144 x 64): M = B * 576 = 2 * B * 288, so the 512-channel layers of a 128-batch are exactly
//     512 tiles = two full rounds of the 256 CUs, the 256-channel layers one round; per unit of K the block stages
//     (288 + 256) * 2 B for 2 * 288 * 256 flop -- 0.65 of the bytes per flop of the 288 x 128 tile and 0.59 of its fragment reads.
//   * FOUR LDS stages of 34 KB (64-B rows, 16-B slots XOR-swizzled by row bits as in conv_igemm_bf16.hip).  K-tile t + 3 is
//     DMA-ed during the load segment of K-tile t into the stage K-tile t - 1 was read from (all its reads were retired by an
//     lgkmcnt(0) in front of an earlier barrier); in the same segment every wave waits with a COUNTED vmcnt for ITS pieces of
//     K-tile t + 1 (two younger K-tiles stay in flight: ~6 barrier intervals between issue and first use), and the first read of
//     K-tile t + 1 happens behind at least one more barrier -- the ordering rule for LDS-DMA data (MI355X_MICROARCH.md).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 288, BN = 256, NW = 8, WM = 2, WN = 4;
constexpr int WTM = BM / WM, WTN = BN / WN;             // 144 x 64
constexpr int TM = WTM / 16, TN = WTN / 16;             // 9 x 4 MFMA tiles per wave
constexpr int BK = 32;                                  // K-tile = one MFMA k-step; LDS rows of 64 B
constexpr int NBUF = 4;
constexpr int GROUPS = (BM + BN) / 16;                  // 34 pieces of 16 rows x 64 B per stage
constexpr int NI = (GROUPS + NW - 1) / NW;              // 5 for waves 0-1, 4 for the others
constexpr int NFULL = GROUPS - (NI - 1) * NW;           // waves below this index own NI pieces, the others NI - 1
constexpr int STAGE = (BM + BN) * 64;                   // 34 816 B
constexpr int SMEM = NBUF * STAGE;

static_assert(SMEM <= 160 * 1024, "the stages must fit the 160 KB LDS");
static_assert(SMEM >= staged_epilogue_smem<BN, TN, WM, NW, 3>(), "the staged epilogue reuses the stage buffers");
static_assert((BM / 16) % 1 == 0 && BM % 16 == 0 && BN % 16 == 0, "pieces cover whole 16-row groups");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// physical 16-B slot of logical k-chunk q in row r: q ^ kSwz[(r >> 2) & 3] (conflict-free ds_read_b128 over 64-B rows)
__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }   // {0, 2, 3, 1}

template <int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 1) igemm_bf16_pp_kernel(const IgemmBfArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;           // wm == wave group (waves w and w + 4 share a SIMD)
    int tile = blockIdx.x;
    if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    const int tile_m = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int taps = p.R * p.S;

    // ---- stager state: piece g = i * NW + wave covers stage rows 16g .. 16g+15; this lane moves physical 16-B slot (lane & 3)
    // of row 16g + (lane >> 2), i.e. logical k-chunk (lane & 3) ^ swz4(row) of that row (the DMA writes lane-linearly)
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[0]), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[0]), 0, p.w_bytes, 0x00020000);
    unsigned vbase[NI], vmask[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = i * NW + wave;
        const int rs = g * 16 + (lane >> 2);
        vbase[i] = 0u;
        vmask[i] = 0u;
        if (g < BM / 16) {                                  // activation rows (wave-uniform branch)
            const int m = m0 + rs;
            const int kq = (lane & 3) ^ swz4(rs);
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
        } else if (g < GROUPS) {                            // weight rows
            const int n = rs - BM;
            const int kq = (lane & 3) ^ swz4(n);
            vbase[i] = (unsigned)(((n0 + n) * p.K + kq * 8) * 2);
            vmask[i] = 0xFFFFFFFFu;
        }
    }
    // K order: 32-channel chunk outer, filter tap inner (the 9 taps of a chunk re-read the same lines of x)
    int tap = 0, c0 = 0, ky = 0, kx = 0;
    if constexpr (DBG & 1024) c0 = ((tile_m * 5 + tile_n * 3) % (p.Cin / BK)) * BK;     // ablation: rotated channel-chunk order per block
    // issue this wave's pieces of the next K-tile in K order into stage `dbuf`; past the end (live = false) the pieces are
    // still issued with every lane out of range (the DMA writes zeros), so that the vmcnt arithmetic stays uniform
    auto issue_tile = [&](int dbuf, bool live) {
        const unsigned soff_a = (unsigned)(((ky * p.Win + kx) * p.Cin + c0) * 2);
        const unsigned soff_b = (unsigned)((tap * p.Cin + c0) * 2);
        const unsigned tbit = live ? (1u << tap) : 0u;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = i * NW + wave;
            if (i == NI - 1 && wave >= NFULL) break;        // ragged last round (wave-uniform)
            const bool is_a = g < BM / 16;                  // wave-uniform
            // the tap / channel-chunk offset goes into the per-lane offset (the range check looks at it alone, and vbase of a
            // border pixel is "negative" until the tap offset is added); masked-out lanes get 0xFFFFFFFF: the DMA writes zeros
            unsigned voff = (vmask[i] & tbit) ? vbase[i] + (is_a ? soff_a : soff_b) : 0xFFFFFFFFu;
            if constexpr (DBG & 64) voff = 0xFFFFFFFFu;            // ablation: every lane out of range (issue + zero fill, no memory traffic)
            if constexpr (DBG & 256) { if (is_a) voff = 0xFFFFFFFFu; }   // ablation: weights only
            if constexpr (DBG & 512) { if (!is_a) voff = 0xFFFFFFFFu; }  // ablation: activations only
            char* dst = smem + dbuf * STAGE + g * 1024;
            if constexpr (!(DBG & 1)) {
                if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 16, voff, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)dst, 16, voff, 0, 0, 0);
            }
        }
        ++tap; ++kx;
        if (kx >= p.S) { kx = 0; ++ky; }
        if (tap >= taps) { tap = 0; kx = 0; ky = 0; c0 += BK; if constexpr (DBG & 1024) { if (c0 >= p.Cin) c0 = 0; } }
    };
    // this wave's pieces of all but the `keep` youngest K-tiles have landed
    auto wait_tiles = [&](auto keep_c) {
        constexpr int KEEP = decltype(keep_c)::value;
        if constexpr ((DBG & 1) || (DBG & 4)) return;
        if (wave < NFULL) wait_vmcnt<KEEP * NI>();
        else wait_vmcnt<KEEP * (NI - 1)>();
    };

    // ---- fragment addressing: v_mfma_f32_16x16x32_bf16 lane l holds A[i = l & 15][k = 8 * (l >> 4) .. +7] (B alike);
    // swz4 of row = base16 + fi does not depend on the 16-row tile index
    const int fi = lane & 15, fq = lane >> 4;
    const int ra = wm * WTM + fi, rb = wn * WTN + fi;
    const int a_off = ra * 64 + ((fq ^ swz4(ra)) << 4);
    const int b_off = BM * 64 + rb * 64 + ((fq ^ swz4(rb)) << 4);
    bf16x8 af[TM], bf[TN];
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = (DBG & 32) ? 0 : p.K / BK;
    // prologue: K-tiles 0 .. NBUF-2 in flight, K-tile 0 landed and visible to everybody
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) issue_tile(t, t < nk);
    wait_tiles(std::integral_constant<int, NBUF - 2>{});
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier behind group 0 from here on

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + buf * STAGE;
        // ---------------- load segment ----------------
        if constexpr (!(DBG & 8)) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 1024);
        }
        issue_tile((buf + NBUF - 1) & (NBUF - 1), kt + NBUF - 1 < nk);     // K-tile kt + 3 into the stage K-tile kt - 1 used
        wait_tiles(std::integral_constant<int, NBUF - 2>{});              // K-tile kt + 1 landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        // ---------------- MFMA segment ----------------
        if constexpr (!(DBG & 128)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (DBG & 16) asm volatile("" :: "v"(af[i]), "v"(bf[j]));
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        if constexpr (!(DBG & 128)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        buf = (buf + 1) & (NBUF - 1);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();              // group 0 waits for group 1's last MFMA segment
    wait_vmcnt<0>();
    __syncthreads();

    igemm_epilogue_staged<BM, BN, TM, TN, WM, NW, 3, true>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
}

template <int DBG>
void launch(const IgemmBfArgs& p, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(igemm_bf16_pp_kernel<DBG>, dim3(blocks), dim3(NW * 64), 0, stream, p);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_conv_igemm_bf16_pp(const IgemmBfArgs& a, hipStream_t stream) {
    // short K (the 1x1 downsample / head convolutions and their dgrads, K = 128 .. 512) included: those launches are bound by their
    // epilogue traffic, and this kernel's 16-byte staged epilogue moves it 2-4x faster than the register-staged kernel's (layer4
    // downsample forward 101 -> 60 us, head conv1 dgrad with the fused BN-backward sums 290 -> 66 us at B = 128)
    static const int min_k = SIMQ_TUNE_INT("SIMQ_BF16_PP_MINK", BK);
    if (a.Cin % BK != 0 || a.Cout % BN != 0 || a.R * a.S > 32 || a.K < min_k) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_BF16_PP", 1);   // 0 = off
    if (mode == 0) return 0;
    int fbm = 0, fbn = 0;
    const bool forced = bf16_forced_tile(a, &fbm, &fbn);
    if (forced && !(fbm == BM && fbn == BN)) return 0;
    const long tilesM = (a.M + BM - 1) / BM;
    const long blocks = tilesM * (a.Cout / BN);
    const long rounds = (blocks + 255) / 256;
    // one block per CU: worth it when the tiles fill (nearly) whole rounds of the 256 CUs
    if (!forced && (blocks < 200 || (double)blocks / (double)(rounds * 256) < 0.85)) return 0;
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = bf16_xcd_chunk((int)blocks, p.tilesN);
    note_launch("igemm_bf16_pp");
    prof_launch_begin(2, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout), stream);
#ifdef SIMQ_ABLATIONS      // timing ablations (tools/pp_check.py): compiled into libsimq_ablate.so only
    static const int dbg = SIMQ_TUNE_INT("SIMQ_BF16_PP_DBG", 0);   // timing ablations (tools/pp_check.py)
    switch (dbg) {
        case 1: launch<1>(p, (unsigned)blocks, stream); break;      // no DMA
        case 2: launch<2>(p, (unsigned)blocks, stream); break;      // no barriers
        case 8: launch<8>(p, (unsigned)blocks, stream); break;      // no fragment reads
        case 16: launch<16>(p, (unsigned)blocks, stream); break;    // no MFMAs
        case 32: launch<32>(p, (unsigned)blocks, stream); break;    // epilogue only
        case 4: launch<4>(p, (unsigned)blocks, stream); break;      // no vmcnt waits
        case 64: launch<64>(p, (unsigned)blocks, stream); break;    // DMA issued with every lane out of range
        case 128: launch<128>(p, (unsigned)blocks, stream); break;  // no s_setprio
        case 256: launch<256>(p, (unsigned)blocks, stream); break;  // weights staged only
        case 512: launch<512>(p, (unsigned)blocks, stream); break;  // activations staged only
        case 1024: launch<1024>(p, (unsigned)blocks, stream); break;  // rotated chunk order
        case 80: launch<80>(p, (unsigned)blocks, stream); break;    // masked DMA, no MFMA
        default: launch<0>(p, (unsigned)blocks, stream);
    }
#else
    launch<0>(p, (unsigned)blocks, stream);
#endif
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
