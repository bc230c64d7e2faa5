// Image-tile 3x3 convolution on the bf16 matrix cores of gfx950 (plain bf16 operands): one block = ONE 24 x 24 feature map
// (576 output pixels) x 128 output channels, the input staged through LDS as a HALO PATCH.
//
// Reference operators: the 3x3 / stride 1 / pad 1 nn.Conv2d forwards of the BasicBlocks (resnet.py:31-47, 23-26) on the 24 x 24
// maps of this network and their dgrads (train.py:132), i.e. the layers conv_igemm_bf16_pp.hip serves.  What that kernel showed
// (tools/pp_check.py ablations, docs/history.md section 4): with the wave groups running one barrier apart the MFMA / fragment-read /
// DMA-issue streams overlap, and what is left is the VOLUME staged through L2 -> LDS: an implicit GEMM re-stages every input
// pixel once per filter tap (9 x), 2.6 GB per layer4 launch at B = 128.  Here, per 32-channel chunk,
//   * the 26 x 26 halo patch of the image (zero border written by the DMA's range check) is staged ONCE -- 43 KB of pixels in a 52-KB
//     buffer -- and the nine taps read their 16-pixel fragments from it at shifted positions (ky: an immediate, kx: one of three base
//     registers); the layout is row-major with an XOR slot swizzle (see the stager): 64-byte DMA requests AND conflict-free ds_read_b128
//     at every tap shift;
//   * only the weights are staged per tap: 128 rows x 64 B = 8 KB per K-tile, exactly one 1-KB DMA piece per wave;
//   -> 12.8 KB staged per K-tile and block instead of 34 KB (per flop: 0.37 x), and a whole image per block halves the number
//      of times the weights are re-read.
//   * 8 waves as 4 x 2 (wave tile 144 x 64, 9 x 4 MFMA tiles = 144 accumulator registers), wave groups {0-3} / {4-7} one barrier
//     apart as in conv_igemm_bf16_pp.hip: load segment (13 fragment reads, this wave's DMA pieces, counted vmcnt, lgkmcnt(0)) |
//     s_barrier | MFMA segment (36 x v_mfma_f32_16x16x32_bf16 under s_setprio 1) | s_barrier.
//   * LDS: two patch buffers of 52 KB (the next chunk's patch is DMA-ed during taps 0..6 of the current one, one 1-KB piece per wave and
//     tap) + a 4-stage ring of 8-KB weight tiles (K-tile t + 3 is DMA-ed during K-tile t).  Ordering of LDS-DMA data: every wave waits for ITS pieces
//     with a counted vmcnt in front of a barrier, and the first read happens behind at least one more barrier.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int HW = 24;                                   // feature map width (halo patch: 26 columns in 32-pixel rows)
constexpr int BN = 128, NW = 8;
constexpr int BK = 32;                                   // K-tile = (32 channels, one tap) = one MFMA k-step
constexpr int TAPS = 9;
constexpr int PPITCH = 32;                               // patch rows are stored 32 pixels apart (26 used): a row = two 1-KB DMA pieces
constexpr int PROW = PPITCH * 64;                        // 2 048 B per patch row: [pixel][4 slots of 16 B = 32 channels]
constexpr int B_STAGES = 4;
constexpr int B_BYTES = BN * 64;                         // 8 192 B per weight tile: one 1-KB piece per wave
static_assert(BN / 16 == NW, "one weight piece per wave and K-tile");

// ROWS = image rows per block: 24 (the whole map, 576 pixels: 4 x 2 waves of 144 x 64) or 12 (half a map, 288 pixels: 2 x 4 waves of
// 144 x 32) -- the half-map form doubles the block count where whole maps x Cout / 128 do not fill the 256 CUs (round 4: the
// 128-channel layers at B = 128, the 256-channel layers at B = 64), at 11 instead of 13 fragment reads per 18 instead of 36 MFMAs.
template <int ROWS>
struct ImgCfg {
    static_assert(ROWS == 24 || ROWS == 12, "whole or half feature maps");
    static constexpr int BM = ROWS * HW;
    static constexpr int WM = ROWS == 24 ? 4 : 2, WN = NW / WM;
    static constexpr int WTM = BM / WM, WTN = BN / WN;       // 144 x 64 | 144 x 32
    static constexpr int TM = WTM / 16, TN = WTN / 16;       // 9 x 4 | 9 x 2 MFMA tiles per wave
    static constexpr int PR = ROWS + 2;                      // patch rows (one halo row above and below)
    static constexpr int A_BYTES = PR * PROW;                // 53 248 | 28 672 B per patch buffer
    static constexpr int PPIECES = 2 * PR;                   // 52 | 28 DMA pieces (16 pixels x 64 B) per 32-channel chunk
    static constexpr int XSLOTS = (PPIECES + NW - 1) / NW;   // patch pieces per wave and channel chunk (7 | 4: pieces q = slot * 8 + wave < PPIECES)
    static constexpr int B_BASE = 2 * A_BYTES;
    static constexpr int OFF_DUMMY = B_BASE + B_STAGES * B_BYTES;   // 1 KB the zero-fill DMAs of empty slots land in
    static constexpr int SMEM_LOOP = OFF_DUMMY + 1024;       // 140 288 | 91 136 B
    static constexpr int SMEM_EPI = staged_epilogue_smem<BN, TN, WM, NW, 3>();
    static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
    static_assert(WTM == 144, "nine 16-pixel tiles per wave");
    static_assert(XSLOTS <= TAPS - 2, "the next chunk's patch is issued during taps 0 .. XSLOTS - 1: the last piece two K-tiles before the chunk ends");
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// 16-byte fragment read through inline assembly (LDS byte address + immediate).  As an ordinary load the compiler would order it
// behind every LDS-DMA in flight once the taps are unrolled (s_waitcnt vmcnt(0) per K-tile: it cannot know that the pieces in flight
// target other stages); the ordering is the kernel's own -- counted vmcnt + s_barrier before a stage is read, lgkmcnt(0) +
// sched_barrier before the fragments are used.
template <int IMM>
__device__ __forceinline__ bf16x8 lds_read16(int addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset field");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
    return v;
}

// DBG (timing ablations, libsimq_ablate.so only; results are wrong by construction): 1 no DMA, 2 no barriers, 8 no fragment reads,
// 16 no MFMAs, 32 epilogue only, 64 DMA issued with every lane out of range, 128 no s_setprio, 256 / 512 no patch / weight pieces,
// 1024 the COST MODEL of a BatchNorm + ReLU applied to the landed patch (round 4: what fusing bn1 into this kernel's operand staging would
// add): in the load segment two taps behind a patch piece's DMA the wave reads its own 1-KB piece back, reads 64 B of per-channel
// coefficients, unpacks / fma / max / packs its 8 values and writes the piece in place -- the real instruction mix at the real point of the
// pipeline, on garbage coefficients
template <int ROWS, int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 2) igemm_bf16_img_kernel(const IgemmBfArgs p) {
    using C = ImgCfg<ROWS>;
    constexpr int BM = C::BM, WM = C::WM, WN = C::WN, WTM = C::WTM, WTN = C::WTN, TM = C::TM, TN = C::TN, A_BYTES = C::A_BYTES,
                  PPIECES = C::PPIECES, XSLOTS = C::XSLOTS, B_BASE = C::B_BASE, OFF_DUMMY = C::OFF_DUMMY, SMEM = C::SMEM;
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int group = wave >> 2;                            // waves w and w + 4 share a SIMD: group 1 runs one barrier behind
    int tile = blockIdx.x;
    if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    const int rb = tile / p.tilesN, tile_n = tile % p.tilesN;              // row block = (image, part of the image)
    const int img = rb / (HW / ROWS), y0 = (rb % (HW / ROWS)) * ROWS;      // first image row of this block
    const int m0 = rb * BM, n0 = tile_n * BN;
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);     // LDS byte address of smem[0]

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[0]), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[0]), 0, p.w_bytes, 0x00020000);

    // ---- patch stager.  The patch lives in LDS row-major, [patch row (pitch 32 pixels)][pixel][4 slots x 16 B]: the 64 bytes of a
    // pixel's 32 channels stay together, so the four lanes of a quad fetch ONE 64-byte run -- a quarter of the L2 requests of a
    // chunk-planar layout, where every 16-byte piece of a pixel travels in a different instruction (measured: with the DMA issued but
    // masked the kernel runs 192 us, with the traffic 257 us, and the L2 request rate -- not its bandwidth -- is what that costs).
    // Pixels 64 B apart collide in LDS banks every four pixels; slot s of the pixel in patch column x therefore holds channel chunk
    // s ^ (3 * ((x >> 2) & 1)): within every 16-lane group a ds_read_b128 serves together, lanes with equal x & 3 sit 4, 8, 12 columns
    // apart and land in four different slots -- conflict-free at every tap shift and across image-row wraps (24 = 0 mod 8).
    // One DMA piece (16 B per lane) = 16 pixels x 64 B = 1 KB = half a patch row; piece q = slot * 8 + wave (q < 52): row q / 2,
    // columns (q & 1) * 16 + (lane >> 2).  Border / padding pixels: byte offset >= 2 GiB (the range check makes the DMA write zeros).
    unsigned abase[XSLOTS];
#pragma unroll
    for (int i = 0; i < XSLOTS; ++i) {
        const int q = i * NW + wave;
        const int py = q >> 1, px = (q & 1) * 16 + (lane >> 2);
        const int iy = y0 + py - 1;                               // image row of patch row py
        const int chunk16 = (lane & 3) ^ (3 * ((px >> 2) & 1));
        const bool ok = q < PPIECES && iy >= 0 && iy < HW && px >= 1 && px <= HW;
        // (invalid pixels: 2 GiB, beyond any tensor this kernel accepts -- adding the chunk offsets below keeps them out of range)
        abase[i] = ok ? (unsigned)((((img * HW + iy) * HW + px - 1) * p.Cin) * 2 + chunk16 * 16) : 0x80000000u;
    }
    const int nchunks = p.Cin / BK;
    auto issue_patch = [&](int slot, int chunk, int abuf) {     // slot: compile-time after unrolling; chunk / abuf: block-uniform
        const int q = slot * NW + wave;
        const bool real = q < PPIECES && chunk < nchunks;
        unsigned voff = real ? abase[slot] + (unsigned)(chunk * BK * 2) : 0xFFFFFFFFu;
        if constexpr (DBG & 64) voff = 0xFFFFFFFFu;
        char* dst = smem + (real ? abuf * A_BYTES + q * 1024 : OFF_DUMMY);
        if constexpr (!(DBG & 1) && !(DBG & 256)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 16, voff, 0, 0, 0);
    };
    // ---- weight stager: this wave's piece = rows 16 * wave .. + 15 of the tile, ROW-major (64 B per row): lane l moves 16-B slot
    // (l & 3) of row (l >> 2), so the four lanes of a quad fetch the 64 contiguous bytes of one row -- one 64-B request instead of
    // four 16-B requests from four different quarter-waves (the L2 request rate, not its bandwidth, is what the staging traffic
    // costs).  Rows of 64 B collide in LDS banks every four rows; slot s of row r therefore holds channel chunk s ^ (3 * ((r >> 2) & 1)):
    // for every 16-lane group a ds_read_b128 serves together the (r & 3, slot) pairs are distinct -- conflict-free.
    const int wrow = lane >> 2, wslot = lane & 3;
    const unsigned wbase = (unsigned)(((n0 + wave * 16 + wrow) * p.K + (wslot ^ (3 * ((wrow >> 2) & 1))) * 8) * 2);
    auto issue_weight = [&](int chunk, int tap, int stage) {       // K-tile (chunk, tap); past the end: zero fill (keeps vmcnt uniform)
        unsigned voff = chunk < nchunks ? wbase + (unsigned)((tap * p.Cin + chunk * BK) * 2) : 0xFFFFFFFFu;
        if constexpr (DBG & 64) voff = 0xFFFFFFFFu;
        char* dst = smem + B_BASE + stage * B_BYTES + wave * 1024;
        if constexpr (!(DBG & 1) && !(DBG & 512)) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)dst, 16, voff, 0, 0, 0);
    };

    // ---- fragment addressing.  v_mfma_f32_16x16x32_bf16 lane l holds A[i = l & 15][k = 8 * (l >> 4) .. +7] (B alike).
    // Output pixel m = wm * 144 + 16 i + fi of the image sits at patch pixel q0 = (m / 24) * 26 + m % 24 (+ tap offset ky * 26 + kx).
    const int fi = lane & 15, fq = lane >> 4;
    // LDS address inside patch buffer 0 for kx = 0, 1, 2 (the swizzle depends on the column); ky is an immediate (one patch row)
    int a_addr[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm * WTM + i * 16 + fi;
        const int y = m / HW, x = m - y * HW;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            a_addr[i][kx] = lds0 + y * PROW + (x + kx) * 64 + ((fq ^ (3 * (((x + kx) >> 2) & 1))) << 4);
    }
    const int b_addr = lds0 + B_BASE + (wn * (WTN / 16)) * 1024 + fi * 64 + ((fq ^ (3 * ((fi >> 2) & 1))) << 4);   // + j * 1024 per 16-row tile (immediate), + stage * B_BYTES
    bf16x8 af[TM], bf[TN];
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk_chunks = (DBG & 32) ? 0 : nchunks;
    // ---- prologue: the patch of chunk 0 and weight tiles 0 .. 2, landed and visible to everybody
#pragma unroll
    for (int sl = 0; sl < XSLOTS; ++sl) issue_patch(sl, 0, 0);
#pragma unroll
    for (int t = 0; t < B_STAGES - 1; ++t) issue_weight(0, t, t);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();

    // One trip = one 32-channel chunk, its nine taps unrolled: K-tile (chunk, tap).  Every LDS address is a base register + an
    // immediate (tap shift, 16-row tile of the weight stage); the only per-K-tile arithmetic is the weight stage (one add) and the
    // DMA source offsets.  Per K-tile and wave: one weight piece (K-tile + 3) and, during taps 0 .. XSLOTS - 1, one piece of the NEXT
    // chunk's patch -- a constant pattern, so the counted wait of every tap is an immediate.
    int kt = 0;
    for (int chunk = 0; chunk < nk_chunks; ++chunk) {
        const int abuf = chunk & 1;
        static_for<TAPS>([&](auto T) {
            constexpr int tap = decltype(T)::value;
            constexpr int toff = (tap / 3) * PROW;                   // ky patch rows down; kx selects the base register
            const int stage = kt & (B_STAGES - 1);
            // ---------------- load segment ----------------
            if constexpr (!(DBG & 8)) {
                const int bs = b_addr + stage * B_BYTES;
                static_for<TN>([&](auto J) { bf[decltype(J)::value] = lds_read16<decltype(J)::value * 1024>(bs); });
                static_for<TM>([&](auto I) { af[decltype(I)::value] = lds_read16<toff>(a_addr[decltype(I)::value][tap % 3]); });
            }
            {
                constexpr int t3 = tap + B_STAGES - 1;           // K-tile three ahead: (chunk, tap + 3) or (chunk + 1, tap - 6)
                issue_weight(t3 < TAPS ? chunk : chunk + 1, t3 < TAPS ? t3 : t3 - TAPS, (stage + B_STAGES - 1) & (B_STAGES - 1));
            }
            constexpr bool patch_tap = tap < XSLOTS;
            if constexpr (patch_tap) issue_patch(tap, chunk + 1, abuf ^ 1);
            if constexpr ((DBG & 1024) != 0 && tap >= 2 && tap - 2 < XSLOTS) {
                constexpr int slot = tap - 2;                      // its DMA was waited for at the end of the previous load segment
                const int q = slot * NW + wave;
                if (q < PPIECES) {                                 // wave-uniform
                    const int pa = lds0 + (abuf ^ 1) * A_BYTES + q * 1024 + lane * 16;
                    const int ca = lds0 + OFF_DUMMY + (lane & 15) * 64;            // (stand-in for a 4-KB coefficient table)
                    bf16x8 v = lds_read16<0>(pa);
                    const bf16x8 c0 = lds_read16<0>(ca), c1 = lds_read16<16>(ca), c2 = lds_read16<32>(ca), c3 = lds_read16<48>(ca);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const floatx4 s0 = __builtin_bit_cast(floatx4, c0), s1 = __builtin_bit_cast(floatx4, c1);
                    const floatx4 h0 = __builtin_bit_cast(floatx4, c2), h1 = __builtin_bit_cast(floatx4, c3);
                    typedef unsigned uintx4r __attribute__((ext_vector_type(4)));
                    const uintx4r raw = __builtin_bit_cast(uintx4r, v);
                    const unsigned rw[4] = {raw[0], raw[1], raw[2], raw[3]};
                    unsigned ow[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lo = __uint_as_float(rw[k] << 16), hi = __uint_as_float(rw[k] & 0xffff0000u);
                        const float a = fmaxf(__builtin_fmaf(lo, k < 2 ? s0[2 * k] : s1[2 * k - 4], k < 2 ? h0[2 * k] : h1[2 * k - 4]), 0.f);
                        const float b = fmaxf(__builtin_fmaf(hi, k < 2 ? s0[2 * k + 1] : s1[2 * k - 3], k < 2 ? h0[2 * k + 1] : h1[2 * k - 3]), 0.f);
                        ow[k] = (unsigned)__builtin_bit_cast(uint16_t, (__bf16)a) | ((unsigned)__builtin_bit_cast(uint16_t, (__bf16)b) << 16);
                    }
                    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                    const uintx4 o = {ow[0], ow[1], ow[2], ow[3]};
                    asm volatile("ds_write_b128 %0, %1" :: "v"(pa), "v"(o) : "memory");
                }
            }
            // counted wait: everything issued two K-tiles ago or earlier has landed = at most the loads of this K-tile and of the one
            // before may be outstanding (one weight piece each, + one patch piece in taps 1 .. XSLOTS)
            constexpr int tp = (tap + TAPS - 1) % TAPS;
            constexpr int outstanding = 2 + (patch_tap ? 1 : 0) + (tp < XSLOTS ? 1 : 0);
            wait_vmcnt<(DBG & (1 | 256 | 512)) ? 0 : outstanding>();
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
            ++kt;
        });
        // the next chunk reads the other patch buffer
        const int delta = abuf ? -A_BYTES : A_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) a_addr[i][kx] += delta;
    }
    if (group == 0) __builtin_amdgcn_s_barrier();           // group 0 waits for group 1's last MFMA segment
    wait_vmcnt<0>();
    __syncthreads();

    igemm_epilogue_staged<BM, BN, TM, TN, WM, NW, 3, true>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
}

template <int DBG>
void launch(const IgemmBfArgs& p, unsigned blocks, int rows, hipStream_t stream) {
    if (rows == 12) hipLaunchKernelGGL((igemm_bf16_img_kernel<12, DBG>), dim3(blocks), dim3(NW * 64), 0, stream, p);
    else hipLaunchKernelGGL((igemm_bf16_img_kernel<24, DBG>), dim3(blocks), dim3(NW * 64), 0, stream, p);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_conv_igemm_bf16_img(const IgemmBfArgs& a, hipStream_t stream) {
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.Hin != HW || a.Win != HW || a.Hout != HW || a.Wout != HW) return 0;
    if (a.Cin % BK != 0 || a.Cout % BN != 0 || a.M % (HW * HW) != 0 || a.x_bytes >= 0x7FFF0000u) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_BF16_IMG", 1);   // 0 = off
    if (mode == 0) return 0;
#ifdef SIMQ_ABLATIONS      // SIMQ_BF16_IMG=2: the four-wave form (conv_igemm_bf16_img4.hip; measured slower, docs/history.md 4) exists in libsimq_ablate.so only
    if (mode == 2) { if (int rc = try_conv_igemm_bf16_img4(a, stream)) return rc; }
#endif
    int fbm = 0, fbn = 0;
    const bool forced = bf16_forced_tile(a, &fbm, &fbn);
    // (forced tiles, tests / tools: "576 x 128" = whole maps; "1288 x 128" = half maps -- 288 x 128 itself names the LDS-DMA kernel's tile)
    constexpr int kForcedHalf = 1288;
    if (forced && !((fbm == HW * HW || fbm == kForcedHalf) && fbn == BN)) return 0;
    // one block per CU: worth it when the tiles fill (nearly) whole rounds of the 256 CUs -- with whole maps, or else with half maps
    // (twice the blocks, wave tile 144 x 32: more fragment reads per MFMA, so only where the whole-map form leaves CUs idle)
    static const int half_mode = SIMQ_TUNE_INT("SIMQ_BF16_IMG_HALF", 1);   // 0: whole maps only (ablation build)
    int rows = 0;
    for (int r : {24, 12}) {
        if (r == 12 && !half_mode) break;
        if (forced && fbm != (r == 24 ? HW * HW : kForcedHalf)) continue;
        const long blk = (long)(a.M / (r * HW)) * (a.Cout / BN);
        const long rnd = (blk + 255) / 256;
        if (forced || (blk >= 200 && (double)blk / (double)(rnd * 256) >= 0.85)) { rows = r; break; }
    }
    if (!rows) return 0;
    const long blocks = (long)(a.M / (rows * HW)) * (a.Cout / BN);
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = bf16_xcd_chunk((int)blocks, p.tilesN);
    note_launch(rows == 24 ? "igemm_bf16_img_whole" : "igemm_bf16_img_half");
    prof_launch_begin(0, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout), stream);
#ifdef SIMQ_ABLATIONS      // timing ablations (tools/pp_check.py): compiled into libsimq_ablate.so only
    static const int dbg = SIMQ_TUNE_INT("SIMQ_BF16_IMG_DBG", 0);   // timing ablations (tools/pp_check.py)
    switch (dbg) {
        case 1: launch<1>(p, (unsigned)blocks, rows, stream); break;      // no DMA
        case 2: launch<2>(p, (unsigned)blocks, rows, stream); break;      // no barriers
        case 8: launch<8>(p, (unsigned)blocks, rows, stream); break;      // no fragment reads
        case 16: launch<16>(p, (unsigned)blocks, rows, stream); break;    // no MFMAs
        case 32: launch<32>(p, (unsigned)blocks, rows, stream); break;    // epilogue only
        case 64: launch<64>(p, (unsigned)blocks, rows, stream); break;    // DMA issued with every lane out of range
        case 80: launch<80>(p, (unsigned)blocks, rows, stream); break;    // masked DMA, no MFMA
        case 17: launch<17>(p, (unsigned)blocks, rows, stream); break;    // no DMA, no MFMA
        case 128: launch<128>(p, (unsigned)blocks, rows, stream); break;  // no s_setprio
        case 256: launch<256>(p, (unsigned)blocks, rows, stream); break;  // no patch pieces (vmcnt(0) waits)
        case 512: launch<512>(p, (unsigned)blocks, rows, stream); break;  // no weight pieces (vmcnt(0) waits)
        case 1024: launch<1024>(p, (unsigned)blocks, rows, stream); break;  // cost model of BatchNorm + ReLU on the landed patch
        default: launch<0>(p, (unsigned)blocks, rows, stream);
    }
#else
    launch<0>(p, (unsigned)blocks, rows, stream);
#endif
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
