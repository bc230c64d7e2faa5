// Image-tile 3x3 convolution on the bf16 matrix cores, FOUR-wave form: the block tile of conv_igemm_bf16_img.hip (one 24 x 24 feature map
// x 128 output channels, halo patch in LDS, weights streamed per tap) with ONE wave per SIMD owning 144 pixels x ALL 128 channels.
//
// Why: conv_igemm_bf16_img.hip's ablations (layer4, B = 128, 348 GFLOP; tools/pp_check.py) -- MFMAs + fragment reads 191 us, MFMAs + DMA
// 202 us, reads + DMA 115 us, all three 246 us -- say the three streams collide in LDS: its 144 x 64 wave tiles read 13 fragments per
// 36 MFMAs, 104 KB per K-tile and CU against the 147 KB the LDS can deliver in the 1152 cycles the MFMAs take, before the DMA writes.
// A 144 x 128 wave tile reads 17 fragments per 72 MFMAs: 68 KB per K-tile and CU (0.65 x).  It needs 288 accumulator registers, i.e.
// one wave per SIMD (512 registers) -- so the overlap the ping-pong wave groups provided has to come from inside the wave:
//   * the reads of K-tile u + 1 are issued between the nine MFMA groups (8 MFMAs: one A fragment x eight B fragments) of K-tile u:
//     B fragments are double-buffered in registers, A fragment i is overwritten by K-tile u + 1's as soon as group i is issued
//     (25 x 4 fragment registers beside the 288 accumulators); inline assembly, so that the compiler neither orders the reads
//     behind the LDS-DMA in flight nor moves them;
//   * one s_barrier per K-tile: in front of it every wave has waited (counted vmcnt) for ITS pieces of K-tile u + 1, behind it the
//     wave issues its DMA pieces of K-tile u + 5 (weights, ring of six 8-KB stages) and of the next channel chunk's patch (thirteen
//     1-KB pieces per wave, three per tap during taps 0..3, one in tap 4) and starts the MFMAs of K-tile u, whose fragments it already holds;
//   * two channel chunks (18 K-tiles) per loop trip, so ring stage, patch buffer and register set of every K-tile are compile-time;
//     DMA source offsets = a per-lane invariant VGPR + the instruction's SCALAR offset; missing pieces go against an empty descriptor.
// Same staging layout, same K order per accumulator and same epilogue as the eight-wave kernel: bit-identical results.
//
// MEASURED AND LEFT OFF (libsimq_ablate.so only, SIMQ_BF16_IMG=2; tools/pp_check.py): layer4 at B = 128 305 us against the eight-wave
// kernel's 246 (layer3 98 | 74); without the DMA 234 (191), without the fragment reads 262, without the MFMAs 150.  A wave issues in
// order: every ds_read_b128 that finds the LDS queue busy -- the four waves leave the barrier together and read at the same instants --
// and every buffer_load ... lds holds back the MFMAs behind it, which the ping-pong form never does (its MFMA segment contains no
// memory instruction).  What the experiment established and the other kernels use or avoid:
//   * the range check of a raw buffer load covers the VGPR offset (+ immediate) only, NOT the scalar offset: a per-lane base that is
//     negative for image 0 (patch row -1) reads zeros although base + scalar offset is a valid address;
//   * MFMA destinations of one function live in ONE register file as far as the compiler is concerned: with 288 accumulators it parks 32
//     in VGPRs and copies them through AGPRs around every use; inline-assembly MFMAs with "+a" / "+v" constraints place them exactly;
//   * an accumulator array whose epilogue unrolls to 9 x 8 x 4 element accesses is not promoted to registers (1152 B of scratch, a
//     store behind every MFMA): the epilogue runs in two 64-channel halves.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int HW = 24, PW = 26;
constexpr int BM = HW * HW, BN = 128, NW = 4, WM = 4;
constexpr int TM = BM / WM / 16, TN = BN / 16;           // 9 x 8 MFMA tiles per wave
constexpr int BK = 32, TAPS = 9;
constexpr int PPITCH = 32, PROW = PPITCH * 64;           // patch rows 32 pixels apart (26 used), 2 048 B per row
constexpr int A_BYTES = PW * PROW;                       // 53 248 B per patch buffer
constexpr int PPIECES = 2 * PW;                          // 52 DMA pieces (16 pixels x 64 B) per 32-channel chunk
constexpr int XSLOTS = PPIECES / NW;                     // 13 per wave
constexpr int B_STAGES = 6, LEAD = B_STAGES - 1;         // weight ring: K-tile u + 5 is DMA-ed during K-tile u
constexpr int TRIP = 2 * TAPS;
constexpr int B_BYTES = BN * 64;                         // 8 192 B per weight tile: two 1-KB pieces per wave
constexpr int WPIECES = BN / 16 / NW;                    // 2
constexpr int B_BASE = 2 * A_BYTES;
constexpr int OFF_DUMMY = B_BASE + B_STAGES * B_BYTES;
constexpr int SMEM_LOOP = OFF_DUMMY + 1024;              // 156 672 B
constexpr int SMEM_EPI = staged_epilogue_smem<BN / 2, TN / 2, WM, NW, 3>();
constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
static_assert(PPIECES % NW == 0 && TRIP % B_STAGES == 0 && SMEM <= 160 * 1024, "layout");

// patch pieces a wave issues in tap t (the next chunk's patch; the last one LEAD - 1 K-tiles before the chunk ends)
constexpr int patch_issues(int tap) { return tap < 4 ? 3 : tap == 4 ? 1 : 0; }
constexpr int patch_first(int tap) { return tap < 4 ? 3 * tap : 12; }
static_assert(patch_first(4) + patch_issues(4) == XSLOTS, "patch pieces per wave");
constexpr int issued(int u) { return WPIECES + patch_issues(((u % TAPS) + TAPS) % TAPS); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// v_mfma_f32_16x16x32_bf16 through inline assembly with the accumulator's register file chosen here: 288 accumulators do not fit the
// 256 AGPRs, and the compiler keeps the MFMA destinations of a function in ONE file -- with the intrinsic it parks 32 accumulators in
// VGPRs and copies them through AGPRs around every use (2 200 v_accvgpr moves per loop trip).  AGPR = true: "a" constraint, else "v".
// Hazards: an accumulator is only ever the srcC / vDst of the same MFMA shape (back-to-back form, no wait states), A / B come from
// LDS behind an s_waitcnt; the epilogue reads the accumulators behind a barrier and explicit s_nops.
template <bool AGPR, bool SKIP = false>
__device__ __forceinline__ void mfma_bf16(floatx4& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (SKIP) asm volatile("" :: "v"(a), "v"(b));          // (timing ablation: fragments consumed, no MFMA)
    else if constexpr (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int IMM>
__device__ __forceinline__ bf16x8 lds_read16(int addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset field");
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
    return v;
}

// DBG (timing ablations, libsimq_ablate.so only; results are wrong by construction): 1 no DMA, 8 no fragment reads, 16 no MFMAs
template <int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 1) igemm_bf16_img4_kernel(const IgemmBfArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    const int img = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = img * BM, n0 = tile_n * BN;
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int nchunks = p.Cin / BK;

    // ---- patch stager (layout of conv_igemm_bf16_img.hip): piece q = slot * 4 + wave: patch row q / 2, columns (q & 1) * 16 + (lane >> 2);
    // slot s of the pixel in column x holds channel chunk s ^ (3 * ((x >> 2) & 1))
    // piece q = slot * 4 + wave sits two patch rows below piece q - 4: ONE per-lane offset (slot 0; lanes on the left / right border
    // carry 2 GiB, out of range for every slot) + a scalar per slot; rows 0 and 25 (slot 0 of waves 0-1, slot 12 of waves 2-3) are
    // wave-uniformly empty and go against the empty descriptor.
    // (the range check of a raw buffer sees the VGPR offset only, so that one has to be a valid offset by itself: the base is the
    // wave's first REAL piece -- slot 1 for waves 0-1, whose slot 0 is patch row 0 -- and the scalar part is never negative)
    const int slot_base = wave < 2 ? 1 : 0;
    unsigned abase0;
    {
        const int py = 2 * slot_base + (wave >> 1), px = (wave & 1) * 16 + (lane >> 2);
        const int chunk16 = (lane & 3) ^ (3 * ((px >> 2) & 1));
        const bool ok = px >= 1 && px <= HW;
        abase0 = ok ? (unsigned)((((img * HW + py - 1) * HW + px - 1) * p.Cin) * 2 + chunk16 * 16) : 0x80000000u;
    }
    const int row2 = 2 * HW * p.Cin * 2;                               // bytes between patch rows py and py + 2
    int oz = 0;          // an opaque scalar zero, refreshed per K-tile: keeps the per-piece scalar offsets from being hoisted out of the trip (SGPR spills)
    auto issue_patch = [&](int slot, int chunk, int abuf) {
        const int py = 2 * slot + (wave >> 1);
        const bool real = chunk < nchunks && py >= 1 && py <= HW;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[0]), 0, real ? p.x_bytes : 0u, 0x00020000);
        char* dst = smem + (chunk < nchunks ? abuf * A_BYTES + (slot * NW + wave) * 1024 : OFF_DUMMY);
        if constexpr (!(DBG & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, abase0, real ? chunk * BK * 2 + (slot - slot_base + oz) * row2 : 0, 0, 0);
    };
    // ---- weight stager: piece h of this wave = rows 16 * (2 wave + h) .. + 15, row-major with the XOR slot swizzle
    const int wrow = lane >> 2, wslot = lane & 3;
    const unsigned wbase = (unsigned)(((n0 + wave * WPIECES * 16 + wrow) * p.K + (wslot ^ (3 * ((wrow >> 2) & 1))) * 8) * 2);
    auto issue_weight = [&](int chunk, int tap, int stage) {
        const bool real = chunk < nchunks;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[0]), 0, real ? p.w_bytes : 0u, 0x00020000);
#pragma unroll
        for (int h = 0; h < WPIECES; ++h) {
            char* dst = smem + B_BASE + stage * B_BYTES + (wave * WPIECES + h) * 1024;
            if constexpr (!(DBG & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, wbase, real ? ((tap + oz) * p.Cin + chunk * BK + (h * 16 + oz) * p.K) * 2 : 0, 0, 0);
        }
    };

    // ---- fragment addressing (as in the eight-wave kernel; this wave owns pixels 144 wave .. + 143 and all eight 16-channel tiles)
    const int fi = lane & 15, fq = lane >> 4;
    int a_addr[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wave * (TM * 16) + i * 16 + fi;
        const int y = m / HW, x = m - y * HW;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            a_addr[i][kx] = lds0 + y * PROW + (x + kx) * 64 + ((fq ^ (3 * (((x + kx) >> 2) & 1))) << 4);
    }
    const int b_addr = lds0 + B_BASE + fi * 64 + ((fq ^ (3 * ((fi >> 2) & 1))) << 4);    // + j * 1024 + stage * B_BYTES (immediates)
    bf16x8 af[TM], bf[2][TN];
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // fragment i / j of K-tile u (trip-relative, may be TRIP: the next trip's first = same immediates as u = 0) into register set u & 1
    auto read_a = [&](auto U, auto I) {
        constexpr int u = decltype(U)::value % TRIP, i = decltype(I)::value, half = u / TAPS, tap = u % TAPS;
        if constexpr (!(DBG & 8)) af[i] = lds_read16<half * A_BYTES + (tap / 3) * PROW>(a_addr[i][tap % 3]);
    };
    auto read_b = [&](auto U, auto J) {
        constexpr int u = decltype(U)::value % TRIP, j = decltype(J)::value;
        if constexpr (!(DBG & 8)) bf[decltype(U)::value & 1][j] = lds_read16<(u % B_STAGES) * B_BYTES + j * 1024>(b_addr);
    };

    // ---- prologue: patch of chunk 0, weight tiles 0 .. LEAD - 1; fragments of K-tile 0
#pragma unroll
    for (int sl = 0; sl < XSLOTS; ++sl) issue_patch(sl, 0, 0);
#pragma unroll
    for (int t = 0; t < LEAD; ++t) issue_weight(0, t, t);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    static_for<TN>([&](auto J) { read_b(std::integral_constant<int, 0>{}, J); });
    static_for<TM>([&](auto I) { read_a(std::integral_constant<int, 0>{}, I); });

    const int nk_chunks = nchunks;
    for (int chunk0 = 0; chunk0 < nk_chunks; chunk0 += 2) {
        static_for<TRIP>([&](auto U) {
            constexpr int u = decltype(U)::value, half = u / TAPS, tap = u % TAPS, set = u & 1;
            using U1 = std::integral_constant<int, u + 1>;
            // my pieces of K-tile u + 1 (weights issued at u + 1 - LEAD, patch at taps <= 4 of the chunk before) have landed: at most
            // the pieces issued during the last LEAD - 2 K-tiles are outstanding; the fragments of K-tile u are in
            wait_vmcnt<(DBG & 1) ? 0 : issued(u - 1) + issued(u - 2) + issued(u - 3)>();
            static_assert(LEAD == 5, "the wait above counts LEAD - 2 = 3 K-tiles");
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");      // everything but A8, issued a moment ago
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
            {
                constexpr int ul = u + LEAD;
                issue_weight(chunk0 + ul / TAPS, ul % TAPS, ul % B_STAGES);
            }
            static_for<patch_issues(tap)>([&](auto Q) { issue_patch(patch_first(tap) + decltype(Q)::value, chunk0 + half + 1, half ^ 1); });
            __builtin_amdgcn_sched_barrier(0);
            // Issue order inside a K-tile: B'0, group 0, A'0, B'1, group 1, A'1, ... B'7, group 7, A'7, group 8, A'8 (X' = fragment of
            // K-tile u + 1; A fragments are single-buffered: A'i replaces Ai as soon as group i's MFMAs are issued, B double-buffered).
            static_for<TM>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (i == TM - 2) asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");   // A8 (issued last in K-tile u - 1) is in
                if constexpr (i < TN) read_b(U1{}, I);
                __builtin_amdgcn_sched_barrier(0);
                static_for<TN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    mfma_bf16<(i < TM - 1), (DBG & 16) != 0>(acc[i][j], af[i], bf[set][j]);   // the last row group's eight accumulators live in VGPRs
                });
                __builtin_amdgcn_sched_barrier(0);
                read_a(U1{}, I);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // (the last MFMAs' results are in the register file)
    __syncthreads();

    // the epilogue in two 64-channel halves (its fully unrolled form over 9 x 8 tiles is not promoted to registers by the compiler: the
    // accumulators would live in scratch memory for the whole kernel)
    static_for<2>([&](auto H) {
        constexpr int h = decltype(H)::value;
        floatx4 half[TM][TN / 2];
        static_for<TM>([&](auto I) { static_for<TN / 2>([&](auto J) {
            half[decltype(I)::value][decltype(J)::value] = acc[decltype(I)::value][h * (TN / 2) + decltype(J)::value]; }); });
        if constexpr (h > 0) __syncthreads();
        igemm_epilogue_staged<BM, BN / 2, TM, TN / 2, WM, NW, 3, true>(p.epi, half, m0, n0 + h * (BN / 2), p.M, p.Cout, smem);
    });
}

template <int DBG>
void launch(const IgemmBfArgs& p, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(igemm_bf16_img4_kernel<DBG>, dim3(blocks), dim3(NW * 64), 0, stream, p);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_conv_igemm_bf16_img4(const IgemmBfArgs& a, hipStream_t stream) {
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.Hin != HW || a.Win != HW || a.Hout != HW || a.Wout != HW) return 0;
    if (a.Cin % (2 * BK) != 0 || a.Cout % BN != 0 || a.M % BM != 0 || a.x_bytes >= 0x7FFF0000u) return 0;
    int fbm = 0, fbn = 0;
    const bool forced = bf16_forced_tile(a, &fbm, &fbn);
    if (forced && !(fbm == BM && fbn == BN)) return 0;
    const long blocks = (long)(a.M / BM) * (a.Cout / BN);
    const long rounds = (blocks + 255) / 256;
    if (!forced && (blocks < 200 || (double)blocks / (double)(rounds * 256) < 0.85)) return 0;
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = bf16_xcd_chunk((int)blocks, p.tilesN);
    note_launch("igemm_bf16_img4");
    prof_launch_begin(0, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout), stream);
#ifdef SIMQ_ABLATIONS
    static const int dbg = SIMQ_TUNE_INT("SIMQ_BF16_IMG_DBG", 0);
    switch (dbg) {
        case 1: launch<1>(p, (unsigned)blocks, stream); break;
        case 8: launch<8>(p, (unsigned)blocks, stream); break;
        case 16: launch<16>(p, (unsigned)blocks, stream); break;
        case 17: launch<17>(p, (unsigned)blocks, stream); break;
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
