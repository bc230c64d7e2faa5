// Weight gradient of the wide 3x3 convolutions on the bf16 matrix cores of gfx950: 256 x 256 (Cout x Cin) tiles per filter tap,
// LDS-DMA staging and wave groups one barrier apart (the structure of conv_igemm_bf16_pp.hip applied to the wgrad contraction).
//
//   dW[co][tap][ci] = sum_p dY[p][co] * X[p (+) tap][ci]        (wgrad half of loss.backward(), train.py:132)
//
// As in conv_wgrad_bf16.hip both operands sit [pixel][channel] in HBM while the contraction runs over pixels, so the tiles are
// staged as stored (rows of 256 channels = 512 B, 32-B chunks XOR-swizzled by pixel-row bits) and read back with
// ds_read_b64_tr_b16 (transpose read: lane t of a 16-lane group addresses 4 channels of pixel row t >> 2 and receives 4 consecutive
// pixels of channel t).  What changes against that kernel:
//   * tile 256 x 256 per tap instead of 128 x 128: 8 waves as 2 x 4, wave tile 128 (co) x 64 (ci) = 32 accumulator tiles, 32 MFMAs per
//     32-pixel step against 24 transpose reads (was 16 against 16) and half the bytes staged per flop;
//   * staging by buffer_load ... lds: one DMA piece = 2 pixel rows x 512 B, 4 pieces per wave and step (2 of dY, 2 of X), the XOR
//     swizzle and the tap shift / image-border test folded into the per-lane SOURCE offset (out-of-image rows: 0xFFFFFFFF -> zeros);
//   * four LDS stages of 32 KB, the pieces of step t + 3 issued during step t, counted vmcnt (8 = two younger steps in flight);
//   * wave groups {0-3} / {4-7} (the two waves of a SIMD) run one barrier apart: load segment | s_barrier | MFMA segment | s_barrier.
// The pixel reduction is split over blocks (fp32 atomics into the zeroed gradient buffer) so that tiles x splits fills two rounds
// of the 256 CUs.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef short short8_ __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int TI = 256, TJ = 256, NW = 8;
constexpr int BRB = 32;                                  // pixels per reduction step (= MFMA k extent)
constexpr int NI = 8, NJ = 4;                            // wave tile 128 (co) x 64 (ci) in 16 x 16 MFMA tiles
constexpr int ROWB = 512;                                // bytes per staged pixel row (256 channels)
constexpr int OP_BYTES = BRB * ROWB;                     // 16 KB per operand and stage
constexpr int STAGE = 2 * OP_BYTES;
constexpr int NBUF = 4;
constexpr int SMEM = NBUF * STAGE;                       // 128 KB
constexpr int HW = 24;

struct WgradPpArgs {
    const uint16_t* x;
    const uint16_t* dy;
    float* dw;
    int Cin, Cout, M, K;
    int tilesI, tilesJ, rows_per_split;
    unsigned x_bytes, dy_bytes;
    int dbg;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// swizzle of the 32-B (16-channel) chunk index by pixel row (16 chunks per row)
__device__ __forceinline__ int rsw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__global__ void __launch_bounds__(NW * 64, 1) wgrad_bf16_pp_kernel(const WgradPpArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 2, wj = wave & 3;             // wi == wave group (waves w and w + 4 share a SIMD)
    int id = blockIdx.x;
    const int tj = id % p.tilesJ; id /= p.tilesJ;
    const int ti = id % p.tilesI;
    const int split = id / p.tilesI;
    const int i0 = ti * TI;
    const int cj_tiles = p.Cin / TJ;
    const int tap = tj / cj_tiles;
    const int cj0 = (tj - tap * cj_tiles) * TJ;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int rbeg = split * p.rows_per_split;
    const int rend = min(p.M, rbeg + p.rows_per_split);
    if (rbeg >= rend) return;                            // block-uniform
    const int nk = (rend - rbeg + BRB - 1) / BRB;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dy), 0, p.dy_bytes, 0x00020000);

    // ---- stager: piece g = i * NW + wave (i = 0, 1: dY rows 2g, 2g + 1; i = 2, 3: X rows 2(g - 16) ...); lane l moves physical
    // 16-B slot (l & 31) of row 2 g' + (l >> 5), i.e. logical 16-B chunk ((slot >> 1) ^ rsw(row)) * 2 + (slot & 1)
    int prow[2];                                         // stage row of this lane for its two pieces of either operand
    unsigned coff[2];                                    // logical channel byte offset inside the 512-B row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int g = i * NW + wave;                     // 0 .. 15
        prow[i] = 2 * g + (lane >> 5);
        const int slot = lane & 31;
        coff[i] = (unsigned)(((((slot >> 1) ^ rsw(prow[i])) << 1) | (slot & 1)) * 16);
    }
    const int shift = (ky - 1) * HW + (kx - 1);          // pixel index shift of this tap inside the [b][24][24] plane
    // running per-piece state, advanced by one 32-pixel step per issue (steps are issued in order 0, 1, 2, ...): the source
    // offsets are linear in the step, the image coordinates of an X row move by (+1 row, +8 columns) -- no division in the loop
    int rr[2], oy[2], ox[2];
    unsigned vy[2], vx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        rr[i] = rbeg + prow[i];
        const int rem = rr[i] % (HW * HW);
        oy[i] = rem / HW; ox[i] = rem - oy[i] * HW;
        vy[i] = (unsigned)(rr[i] * p.Cout + i0) * 2u + coff[i];
        vx[i] = (unsigned)((rr[i] + shift) * p.Cin + cj0) * 2u + coff[i];
    }
    const unsigned step_y = (unsigned)(BRB * p.Cout * 2), step_x = (unsigned)(BRB * p.Cin * 2);
    auto issue_step = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {                    // dY pieces
            const unsigned voff = rr[i] < rend ? vy[i] : 0xFFFFFFFFu;
            char* dst = smem + stage * STAGE + (i * NW + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void*)dst, 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                    // X pieces: pixel r shifted by the tap, zero outside the image
            const bool ok = rr[i] < rend && (unsigned)(oy[i] + ky - 1) < (unsigned)HW && (unsigned)(ox[i] + kx - 1) < (unsigned)HW;
            const unsigned voff = ok ? vx[i] : 0xFFFFFFFFu;
            char* dst = smem + stage * STAGE + OP_BYTES + (i * NW + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                    // advance to the next step: 32 pixels = 1 image row + 8 columns
            rr[i] += BRB; vy[i] += step_y; vx[i] += step_x;
            ox[i] += 8; oy[i] += 1;
            if (ox[i] >= HW) { ox[i] -= HW; oy[i] += 1; }
            if (oy[i] >= HW) oy[i] -= HW;
        }
    };

    // ---- transpose-read fragments: lane (t = lane & 15, g = lane >> 4) -> 8 pixels k = 8g .. 8g+7 of channel (chunk * 16 + t)
    const int ft = lane & 15, fg = lane >> 4;
    const int r0 = 8 * fg + (ft >> 2), r1 = r0 + 4;
    int a_off[NI][2], b_off[NJ][2];                      // byte offsets inside a stage
#pragma unroll
    for (int a = 0; a < NI; ++a) {
        const int chunk = wi * NI + a;
        a_off[a][0] = r0 * ROWB + ((chunk ^ rsw(r0)) << 5) + ((ft & 3) << 3);
        a_off[a][1] = r1 * ROWB + ((chunk ^ rsw(r1)) << 5) + ((ft & 3) << 3);
    }
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
        const int chunk = wj * NJ + b;
        b_off[b][0] = OP_BYTES + r0 * ROWB + ((chunk ^ rsw(r0)) << 5) + ((ft & 3) << 3);
        b_off[b][1] = OP_BYTES + r1 * ROWB + ((chunk ^ rsw(r1)) << 5) + ((ft & 3) << 3);
    }
    auto frag = [&](const char* st, const int (&off)[2]) -> bf16x8 {
        short4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_ __attribute__((address_space(3)))*)(st + off[0]));
        short4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_ __attribute__((address_space(3)))*)(st + off[1]));
        short8_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    bf16x8 af[NI], bf[NJ];
    floatx4 acc[NI][NJ];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < NJ; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

    // prologue: steps 0 .. 2 in flight, step 0 landed and visible to everybody
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) issue_step(t);
    wait_vmcnt<(NBUF - 2) * 4>();
    __builtin_amdgcn_s_barrier();
    if (wi == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0 from here on

    for (int kt = 0; kt < nk; ++kt) {
        const int stage = kt & (NBUF - 1);
        const char* st = smem + stage * STAGE;
        // ---------------- load segment ----------------
#pragma unroll
        for (int b = 0; b < NJ; ++b) bf[b] = frag(st, b_off[b]);
#pragma unroll
        for (int a = 0; a < NI; ++a) af[a] = frag(st, a_off[a]);
        issue_step((stage + NBUF - 1) & (NBUF - 1));                    // step kt + 3 into the stage step kt - 1 used
        wait_vmcnt<(NBUF - 2) * 4>();                                    // step kt + 1 landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- MFMA segment ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < NJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wi == 0) __builtin_amdgcn_s_barrier();           // group 0 waits for group 1's last MFMA segment
    wait_vmcnt<0>();

    if (p.dbg & 1) {                                     // timing ablation: no atomics
        if (acc[0][0][0] == 12345.678f) p.dw[0] = 1.f;
        return;
    }
    // C/D layout: col = lane & 15 (-> ci), row = 4 * (lane >> 4) + reg (-> co)
#pragma unroll
    for (int b = 0; b < NJ; ++b) {
        const size_t col = (size_t)tap * p.Cin + cj0 + wj * (TJ / 4) + b * 16 + ft;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * (TI / 2) + a * 16 + 4 * fg + r;
                unsafeAtomicAdd(p.dw + (size_t)i * p.K + col, acc[a][b][r]);
            }
        }
    }
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered (caller: conv_wgrad_bf16.hip), < 0 on error
int try_conv_wgrad_bf16_pp(const uint16_t* x, const uint16_t* dy, float* dw, const ConvGeom& g, unsigned x_bytes, unsigned dy_bytes,
                           hipStream_t stream) {
    if (g.R != 3 || g.S != 3 || g.stride != 1 || g.pad != 1 || g.Hin != HW || g.Win != HW || g.Hout != HW || g.Wout != HW) return 0;
    if (g.Cin % TJ != 0 || g.Cout % TI != 0) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_BF16_WGRAD_PP", 1);   // 0 = off
    if (mode == 0) return 0;
    WgradPpArgs p;
    p.x = x; p.dy = dy; p.dw = dw; p.Cin = g.Cin; p.Cout = g.Cout; p.M = g.M(); p.K = g.K();
    p.x_bytes = x_bytes; p.dy_bytes = dy_bytes;
    p.tilesI = g.Cout / TI;
    p.tilesJ = 9 * (g.Cin / TJ);
    const int tiles = p.tilesI * p.tilesJ;
    const int rsteps = (p.M + BRB - 1) / BRB;
    if (rsteps < 64) return 0;                           // tiny batches: the register-staged kernel
    // splits: tiles x splits fills whole rounds of the 256 CUs (one block per CU), at least ~24 steps per block
    static const int dbg = SIMQ_TUNE_INT("SIMQ_BF16_WGRAD_PP_DBG", 0);
    p.dbg = dbg;
    int best_s = 1;
    double best = 1e300;
    for (int s = 1; s <= rsteps / 24 && s <= 64; ++s) {
        const long rounds = ((long)tiles * s + 255) / 256;
        const double cost = (double)rounds * ((rsteps + s - 1) / s + 6);
        if (cost < best * 0.999) { best = cost; best_s = s; }
    }
    if (dbg >> 4) best_s = dbg >> 4;                     // timing ablation: forced split count
    int rps = (p.M + best_s - 1) / best_s;
    rps = ((rps + BRB - 1) / BRB) * BRB;
    const int splits = (p.M + rps - 1) / rps;
    p.rows_per_split = rps;
    note_launch("wgrad_bf16_pp");
    prof_launch_begin(1, 2.0 * p.M * p.Cout * p.K, 2.0 * ((double)p.M * (g.Cin + g.Cout)) + 4.0 * (double)p.Cout * p.K, stream);
    hipLaunchKernelGGL(wgrad_bf16_pp_kernel, dim3((unsigned)(tiles * splits)), dim3(NW * 64), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
