// Image-tile 3x3 convolution on the bf16 matrix cores of gfx950 (plain bf16 operands): one block = ONE 24 x 24 feature map
// (576 output pixels) x 128 output channels, the input staged through LDS as a HALO PATCH.
//
// Reference operators: the 3x3 / stride 1 / pad 1 nn.Conv2d forwards of the BasicBlocks (resnet.py:31-47, 23-26) on the 24 x 24
// maps of this network and their dgrads (train.py:132), i.e. the layers conv_igemm_bf16_pp.hip serves.  What that kernel showed
// (tools/pp_check.py ablations): with the wave groups running one barrier apart the MFMA / fragment-read /
// DMA-issue streams overlap, and what is left is the VOLUME staged through L2 -> LDS: an implicit GEMM re-stages every input
// pixel once per filter tap (9 x), 2.6 GB per layer4 launch at B = 128.  Here, per 32-channel chunk,
//   * the 26 x 26 halo patch of the image (zero border written by the DMA's range check) is staged ONCE -- 43 KB -- and the
//     nine taps read their 16-pixel fragments from it at shifted positions: the pixel dimension of the patch is laid out
//     16 pixels x 16 B per 256-B line ("chunk-major" pieces), so ANY run of 16 consecutive patch pixels covers all 64 banks:
//     conflict-free ds_read_b128 at every tap shift without a swizzle;
//   * only the weights are staged per tap: 128 rows x 64 B = 8 KB per K-tile, exactly one 1-KB DMA piece per wave;
//   -> 12.8 KB staged per K-tile and block instead of 34 KB (per flop: 0.37 x), and a whole image per block halves the number
//      of times the weights are re-read.
//   * 8 waves as 4 x 2 (wave tile 144 x 64, 9 x 4 MFMA tiles = 144 accumulator registers), wave groups {0-3} / {4-7} one barrier
//     apart as in conv_igemm_bf16_pp.hip: load segment (13 fragment reads, this wave's DMA pieces, counted vmcnt, lgkmcnt(0)) |
//     s_barrier | MFMA segment (36 x v_mfma_f32_16x16x32_bf16 under s_setprio 1) | s_barrier.
//   * LDS: two patch buffers of 43 KB (the next chunk's patch is DMA-ed during taps 0..5 of the current one) + a 4-stage ring of
//     8-KB weight tiles (K-tile t + 3 is DMA-ed during K-tile t).  Ordering of LDS-DMA data: every wave waits for ITS pieces
//     with a counted vmcnt in front of a barrier, and the first read happens behind at least one more barrier.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int HW = 24, PW = 26;                          // feature map / halo patch width
constexpr int BM = HW * HW, BN = 128, NW = 8, WM = 4, WN = 2;
constexpr int WTM = BM / WM, WTN = BN / WN;              // 144 x 64
constexpr int TM = WTM / 16, TN = WTN / 16;              // 9 x 4 MFMA tiles per wave
constexpr int BK = 32;                                   // K-tile = (32 channels, one tap) = one MFMA k-step
constexpr int TAPS = 9;
constexpr int PATCH_PX = PW * PW;                        // 676 patch pixels
constexpr int A_GROUPS = (PATCH_PX + 15) / 16;           // 43 groups of 16 patch pixels
constexpr int PLANE = A_GROUPS * 256;                    // one 16-B channel chunk of every patch pixel: 11 008 B (a multiple of 256)
constexpr int A_BYTES = 4 * PLANE;                       // 44 032 B per patch buffer: [chunk 0..3][pixel][16 B]
constexpr int NA = (A_GROUPS + NW - 1) / NW;             // <= 6 pixel groups per wave
constexpr int NPD = 4 * NA;                              // <= 24 patch DMA instructions (256 B each) per wave and channel chunk
constexpr int B_STAGES = 4;
constexpr int B_BYTES = BN * 64;                         // 8 192 B per weight tile: one 1-KB piece per wave
constexpr int B_BASE = 2 * A_BYTES;
constexpr int SMEM_LOOP = B_BASE + B_STAGES * B_BYTES;   // 120 832 B
constexpr int SMEM_EPI = staged_epilogue_smem<BN, TN, WM, NW, 3>();
constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;

static_assert(BN / 16 == NW, "one weight piece per wave and K-tile");
static_assert(NA <= TAPS - 2, "the next chunk's patch is issued during the first taps and waited for before the chunk ends");
static_assert(SMEM <= 160 * 1024, "LDS budget");
static_assert(PLANE % 256 == 0, "planes start on a bank-0 boundary: 16 consecutive pixels of a plane cover all 64 banks");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 1) igemm_bf16_img_kernel(const IgemmBfArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int group = wave >> 2;                            // waves w and w + 4 share a SIMD: group 1 runs one barrier behind
    int tile = blockIdx.x;
    if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    const int img = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = img * BM, n0 = tile_n * BN;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[0]), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[0]), 0, p.w_bytes, 0x00020000);

    // ---- patch stager.  The patch lives in LDS as four planes [16-B channel chunk][patch pixel][16 B]: a fragment's 16 pixels
    // are 256 contiguous bytes and its address is AFFINE in the pixel index, so the nine taps differ by an immediate offset.
    // One DMA instruction (dword per lane) moves 16 pixels of one chunk: lane l carries word (l & 3) of pixel 16 g + (l >> 2);
    // the DMA writes lane-linearly (dst + 4 l), which is exactly pixel-major within the plane.  Wave w owns pixel groups
    // g = i * NW + w.  Border / padding pixels: byte offset 0xFFFFFFFF (the range check makes the DMA write zeros).
    unsigned abase[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = (i * NW + wave) * 16 + (lane >> 2);
        const int py = q / PW, px = q - py * PW;
        const bool ok = q < PATCH_PX && py >= 1 && py <= HW && px >= 1 && px <= HW;
        // (invalid pixels: 2 GiB, beyond any plane this kernel accepts -- adding the chunk offsets below keeps them out of range)
        abase[i] = ok ? (unsigned)((((img * HW + py - 1) * HW + px - 1) * p.Cin) * 2 + (lane & 3) * 4) : 0x80000000u;
    }
    const int na_mine = (A_GROUPS - wave + NW - 1) / NW;      // pixel groups this wave owns (6 for waves 0-2, 5 for the others)
    const int npd_mine = 4 * na_mine;
    // patch DMA d (0 .. npd_mine-1) of the chunk with channel offset cbytes: pixel group d / 4, channel chunk d % 4
    auto issue_patch = [&](auto d_c, unsigned cbytes, int abuf) {
        constexpr int D = decltype(d_c)::value;
        constexpr int I = D / 4, C = D % 4;
        unsigned voff = abase[I] + (cbytes + C * 16);          // one VALU add of a scalar; nothing per-(group, chunk) is kept in registers
        if constexpr (DBG & 64) voff = 0xFFFFFFFFu;
        char* dst = smem + abuf * A_BYTES + C * PLANE + (I * NW + wave) * 256;
        if constexpr (!(DBG & 1) && !(DBG & 256)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 4, voff, 0, 0, 0);
    };
    // ---- weight stager: this wave's piece = rows 16 * wave .. + 15 of the tile, chunk-major (16 rows x 16 B per 256-B line):
    // lane l moves row (l & 15), chunk (l >> 4) and the DMA's lane-linear write puts it at (l >> 4) * 256 + (l & 15) * 16
    const unsigned wbase = (unsigned)(((n0 + wave * 16 + (lane & 15)) * p.K + (lane >> 4) * 8) * 2);
    const int nchunks = p.Cin / BK;
    auto issue_weight = [&](int chunk, int tap, int stage) {       // K-tile (chunk, tap); past the end: zero fill (keeps vmcnt uniform)
        unsigned voff = chunk < nchunks ? wbase + (unsigned)((tap * p.Cin + chunk * BK) * 2) : 0xFFFFFFFFu;
        if constexpr (DBG & 64) voff = 0xFFFFFFFFu;
        char* dst = smem + B_BASE + stage * B_BYTES + wave * 1024;
        if constexpr (!(DBG & 1) && !(DBG & 512)) __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)dst, 16, voff, 0, 0, 0);
    };

    // ---- fragment addressing.  v_mfma_f32_16x16x32_bf16 lane l holds A[i = l & 15][k = 8 * (l >> 4) .. +7] (B alike).
    // Output pixel m = wm * 144 + 16 i + fi of the image sits at patch pixel q0 = (m / 24) * 26 + m % 24 (+ tap offset ky * 26 + kx).
    const int fi = lane & 15, fq = lane >> 4;
    int a_addr[TM];                                          // byte offset inside a patch buffer, tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm * WTM + i * 16 + fi;
        a_addr[i] = fq * PLANE + ((m / HW) * PW + (m % HW)) * 16;
    }
    const int b_addr = B_BASE + (wn * (WTN / 16)) * 1024 + fq * 256 + fi * 16;      // + j * 1024 per 16-row tile, + stage * B_BYTES
    bf16x8 af[TM], bf[TN];
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk_chunks = (DBG & 32) ? 0 : nchunks;
    // ---- prologue: the patch of chunk 0 and weight tiles 0 .. 2 in flight; patch + tile 0 landed and visible to everybody
    {
        auto all = [&](auto self, auto d_c) {
            constexpr int D = decltype(d_c)::value;
            if constexpr (D < NPD) {
                if (D < npd_mine) issue_patch(d_c, 0u, 0);
                self(self, std::integral_constant<int, D + 1>{});
            }
        };
        all(all, std::integral_constant<int, 0>{});
    }
#pragma unroll
    for (int t = 0; t < B_STAGES - 1; ++t) issue_weight(0, t, t);
    wait_vmcnt<B_STAGES - 2>();
    __builtin_amdgcn_s_barrier();
    if (group == 1) __builtin_amdgcn_s_barrier();

    // One iteration = one K-tile (chunk, tap).  The tap loop is NOT unrolled (an unrolled body made the compiler keep per-tap
    // addresses live and spill -- and a scratch reload costs an s_waitcnt vmcnt(0), which drains the DMA queue): the tap offset is a
    // scalar added to the 9 fragment addresses (9 VALU per load segment).  Measured alternatives on layer4 at B = 128 (this form:
    // 287-296 us): the NEXT segment's address arithmetic between the MFMAs of the wave's own MFMA segment -- 311 us (the MFMA
    // segment grows by more than the load segment shrinks); the same arithmetic behind the fragment reads / DMA issue of the load
    // segment (in the shadow of the LDS latency) -- 308 us, and with the DMA source steps in the instructions' scalar offsets 317 us
    // (each variant is FASTER without the DMA traffic -- 203 vs 208 us -- so what they lose is the timing of the DMA issue relative
    // to the partner wave's MFMA segment, not instruction count); taps unrolled with immediate offsets -- the compiler spills.
    int chunk = 0, tap = 0, toff = 0, tx = 0, prev_issued = 0;
    const int nk = nk_chunks * TAPS;
    for (int kt = 0; kt < nk; ++kt) {
        const int abuf = chunk & 1;
        const int stage = kt & (B_STAGES - 1);
        // ---------------- load segment ----------------
        if constexpr (!(DBG & 8)) {
            const char* bs = smem + stage * B_BYTES + b_addr;
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(bs + j * 1024);
            const char* as = smem + abuf * A_BYTES + toff;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(as + a_addr[i]);
        }
        // staging: during tap t < na_mine the four channel-chunk planes of pixel group t of the NEXT chunk's patch, then this wave's
        // weight piece of the K-tile three ahead
        int issued = 0;
        if (chunk + 1 < nchunks && tap < na_mine) {
            unsigned ab = abase[0];                          // abase[tap]: select among the NA registers (tap is wave-uniform)
#pragma unroll
            for (int i = 1; i < NA; ++i) ab = tap == i ? abase[i] : ab;
            unsigned voff = ab + (unsigned)((chunk + 1) * BK * 2);
            if constexpr (DBG & 64) voff = 0xFFFFFFFFu;
            char* dst = smem + (abuf ^ 1) * A_BYTES + (tap * NW + wave) * 256;
            if constexpr (!(DBG & 1) && !(DBG & 256)) {
                // (the 16-B channel-chunk step rides in the instruction's scalar offset: it is added to the memory address only)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(dst), 4, voff, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(dst + PLANE), 4, voff, 16, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(dst + 2 * PLANE), 4, voff, 32, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(dst + 3 * PLANE), 4, voff, 48, 0, 0);
            }
            issued = 4;
        }
        {
            const int t3 = tap + B_STAGES - 1;               // K-tile three ahead: (chunk, tap + 3) or (chunk + 1, tap - 6)
            issue_weight(t3 < TAPS ? chunk : chunk + 1, t3 < TAPS ? t3 : t3 - TAPS, (stage + B_STAGES - 1) & (B_STAGES - 1));
        }
        // the weight piece of the NEXT K-tile (issued two load segments ago) has landed once at most the operations issued since
        // then are outstanding: 2 weight pieces + the patch DMAs of this and of the previous load segment (0, 4 or 8)
        if constexpr (!(DBG & 1) && !(DBG & 4) && !(DBG & 256) && !(DBG & 512)) {
            const int extra = issued + prev_issued;
            if (extra == 0) wait_vmcnt<2>();
            else if (extra == 4) wait_vmcnt<6>();
            else wait_vmcnt<10>();
        }
        prev_issued = issued;
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
        // next (chunk, tap); toff = ((tap / 3) * 26 + tap % 3) * 16
        ++tap; ++tx; toff += 16;
        if (tx == 3) { tx = 0; toff += (PW - 3) * 16; }
        if (tap == TAPS) { tap = 0; tx = 0; toff = 0; ++chunk; }
    }
    if (group == 0) __builtin_amdgcn_s_barrier();           // group 0 waits for group 1's last MFMA segment
    wait_vmcnt<0>();
    __syncthreads();

    igemm_epilogue_staged<BM, BN, TM, TN, WM, NW, 3, true>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
}

template <int DBG>
void launch(const IgemmBfArgs& p, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(igemm_bf16_img_kernel<DBG>, dim3(blocks), dim3(NW * 64), 0, stream, p);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_conv_igemm_bf16_img(const IgemmBfArgs& a, hipStream_t stream) {
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.Hin != HW || a.Win != HW || a.Hout != HW || a.Wout != HW) return 0;
    if (a.Cin % BK != 0 || a.Cout % BN != 0 || a.M % BM != 0 || a.x_bytes >= 0x7FFF0000u) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_BF16_IMG", 1);   // 0 = off
    if (mode == 0) return 0;
    int fbm = 0, fbn = 0;
    const bool forced = tune_forced_tile(&fbm, &fbn);
    if (forced && !(fbm == BM && fbn == BN)) return 0;
    const long blocks = (long)(a.M / BM) * (a.Cout / BN);
    const long rounds = (blocks + 255) / 256;
    // one block per CU: worth it when the tiles fill (nearly) whole rounds of the 256 CUs
    if (!forced && (blocks < 200 || (double)blocks / (double)(rounds * 256) < 0.85)) return 0;
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = bf16_xcd_chunk((int)blocks, p.tilesN);
    prof_launch_begin(0, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout), stream);
#ifdef SIMQ_ABLATIONS      // timing ablations (tools/pp_check.py): compiled into libsimq_ablate.so only
    static const int dbg = SIMQ_TUNE_INT("SIMQ_BF16_IMG_DBG", 0);   // timing ablations (tools/pp_check.py)
    switch (dbg) {
        case 1: launch<1>(p, (unsigned)blocks, stream); break;      // no DMA
        case 2: launch<2>(p, (unsigned)blocks, stream); break;      // no barriers
        case 4: launch<4>(p, (unsigned)blocks, stream); break;      // no vmcnt waits
        case 8: launch<8>(p, (unsigned)blocks, stream); break;      // no fragment reads
        case 16: launch<16>(p, (unsigned)blocks, stream); break;    // no MFMAs
        case 32: launch<32>(p, (unsigned)blocks, stream); break;    // epilogue only
        case 64: launch<64>(p, (unsigned)blocks, stream); break;    // DMA issued with every lane out of range
        case 80: launch<80>(p, (unsigned)blocks, stream); break;    // masked DMA, no MFMA
        case 17: launch<17>(p, (unsigned)blocks, stream); break;    // no DMA, no MFMA
        case 256: launch<256>(p, (unsigned)blocks, stream); break;  // no patch pieces (no vmcnt waits)
        case 512: launch<512>(p, (unsigned)blocks, stream); break;  // no weight pieces (no vmcnt waits)
        case 25: launch<25>(p, (unsigned)blocks, stream); break;    // barriers only
        case 128: launch<128>(p, (unsigned)blocks, stream); break;  // no s_setprio
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
