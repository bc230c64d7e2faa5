// Image-tile 3x3 convolution in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32) for the 64-input-channel layers of the
// encoder: one block = THREE image rows (72 output pixels) x 64 output channels, the input staged once as a halo patch.
//
// Reference operators: the 3x3 / stride 1 / pad 1 nn.Conv2d forwards of layer1's BasicBlocks and of layer2's first convolution
// (resnet.py:31-47 on the 24 x 24 maps, 64 input channels) and the dgrads (train.py:132) that contract over 64 channels.
//
// Why: with N = 64 the implicit GEMM (conv_igemm.hip) needs 32x32 tiles to fill the 256 CUs at B = 32 (M = 18 432), and a 32x32
// tile re-fetches 73 KB of im2col rows and 73 KB of weights for 2.4 MFLOP: 170 MB through L2 -> LDS per launch, 29 us = 45 TF/s for a
// 1.36-GFLOP problem.  Here, per block,
//   * the 5 x 26 halo patch of the three rows (all 64 channels, 35 KB; zero border from the buffer range check) is DMA-ed ONCE and
//     the nine taps read their fragments from it at shifted positions (an immediate per tap and 16-channel chunk);
//   * the weights stream per tap: [4 chunks of 16 channels][64 rows][64 B] = 16 KB, two stages, two 1-KB DMA pieces per wave and tap,
//     issued one tap ahead (three stages / two taps ahead measured the same at B = 32 and cost the second resident block);
//   -> 35 + 147 KB staged per 72 x 64 x 576 tile (0.25 x), B * 8 blocks = exactly one per CU at B = 32.
//   * 8 waves = 2 (pixel halves: tiles 0-2 / 3-4 of the five 16-pixel tiles) x 4 (16 output channels each); the two waves of a SIMD
//     hold five tiles together.  Fragments through inline-assembly ds_read_b128 one K-step ahead (the compiler would order ordinary
//     LDS loads behind every DMA in flight), counted vmcnt + one s_barrier per tap.
// LDS: patch [pixel][64 floats + 16 B pad] (consecutive pixels 4 banks apart: a 16-lane group of a ds_read_b128 covers all 64 banks;
// the tiles that wrap an image row lose two lanes to conflicts); weight rows of 64 B with the 16-B slots XOR-swizzled by row bits.
// Measured alone (tools/imgf32_check.py, us per launch, image-tile | best implicit-GEMM tile): B = 32 64->64 22.7 | 27.6, 64->128 32.7 | 35.8;
// B = 29 20.6 | 26.6 and 30.0 | 34.6; B = 128 63.8 | 59.5 and 107.8 | 95.8 -- with many rounds of blocks the 64x32 / 64x64 tiles
// amortise their re-fetches and win, so the launcher takes this kernel for up to two blocks per CU only.
// On the whole fp32 step (tools/ab_step.py, six alternating pairs): 3211.8 tr/s without, 3216.7 with this kernel -- neutral: in the step its
// launches share the device with the side stream's forward, and 512-thread blocks with 68 KB of LDS find fewer free CUs than 256-thread
// blocks with 8 KB (40 us per launch in the step's trace against 33 us for the 32x32 tile).
// Arithmetic: exact fp32 FMA chains as in conv_igemm.hip (the fp32 parity bars apply unchanged); K order (tap, channel) instead of
// (channel chunk, tap).  Epilogue: igemm_epilogue.h (bias, BN statistics, folded BN, residual, ReLU, fused BN-backward sums).
#include <cstdint>
#include <type_traits>

#include "common.h"
#include "bn_coeff.h"
#include "igemm_epilogue.h"

namespace simq {

namespace {

typedef __attribute__((address_space(3))) void lds_void;

constexpr int HW = 24, PW = 26, ROWS = 3, PR = ROWS + 2;
constexpr int CIN = 64, BN = 64, NW = 8, TAPS = 9, NCC = CIN / 16;
constexpr int BMR = ROWS * HW;                           // 72 output pixels per block ...
constexpr int BM = 96, TM = 3;                           // ... in 2 wave rows x 3 tiles of 16 (tile 5 is empty, tile 4 half)
constexpr int PPITCH = CIN * 4 + 16;                     // 272 B per patch pixel
constexpr int PATCH_BYTES = PR * PW * PPITCH;            // 35 360
constexpr int PATCH_PIECES = (PATCH_BYTES + 1023) / 1024;   // 35 DMA pieces of 1 KB
constexpr int XSLOTS = (PATCH_PIECES + NW - 1) / NW;     // 5 per wave
constexpr int W_BASE = PATCH_PIECES * 1024;
constexpr int W_STAGE = BN * CIN * 4;                    // 16 384 B: one tap
constexpr int W_STAGES = 2;                              // 68.6 KB in all: two blocks per CU
constexpr int LEAD = W_STAGES - 1;                       // taps between a weight DMA and its first read
constexpr int W_PIECES = W_STAGE / 1024 / NW;            // 2 per wave and tap
constexpr int SMEM = W_BASE + W_STAGES * W_STAGE;        // 68 608 B
static_assert(W_PIECES * NW * 1024 == W_STAGE, "weight pieces");

struct ImgF32Args {
    const float* x; const float* w;
    EpiArgs epi;
    int M, Cout, tilesN;
    unsigned x_bytes, w_bytes;
    InBn in;   // XBN: x is a pre-BatchNorm output; the patch becomes relu(x * scale[ci] + shift[ci]) in LDS (common.h InBn)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    static_assert(N >= 0 && N < 16, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory");
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
template <int IMM>
__device__ __forceinline__ floatx4 lds_read16(int addr) {
    static_assert(IMM >= 0 && IMM < 65536 && IMM % 16 == 0, "ds_read_b128 offset field");
    floatx4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
    return v;
}

template <bool XBN>
__global__ void __launch_bounds__(NW * 64, 2) conv_img_f32_kernel(const ImgF32Args p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // waves w and w + 4 share a SIMD: pixel halves of the same 16 channels
    const int tile = blockIdx.x;
    const int rb = tile / p.tilesN, tn = tile - rb * p.tilesN;
    const int img = rb / (HW / ROWS), y0 = (rb - img * (HW / ROWS)) * ROWS;
    const int m0 = (img * HW + y0) * HW, n0 = tn * BN;
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    // ---- patch: piece q = 1 KB of the [pixel][272 B] array; lane l moves its bytes 16 l .. 16 l + 15
    {
        __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
#pragma unroll
        for (int s = 0; s < XSLOTS; ++s) {
            const int q = s * NW + wave;
            if (q < PATCH_PIECES) {                                                 // wave-uniform
                const int off = q * 1024 + lane * 16;
                const int pix = off / PPITCH, slot = (off - pix * PPITCH) >> 4;     // slot 16 = the pad
                const int prow = pix / PW, pcol = pix - prow * PW;
                const int iy = y0 - 1 + prow, ix = pcol - 1;
                const bool ok = pix < PR * PW && slot < 16 && (unsigned)iy < (unsigned)HW && (unsigned)ix < (unsigned)HW;
                const unsigned voff = ok ? (unsigned)((((img * HW + iy) * HW + ix) * CIN + slot * 4) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(smem + q * 1024), 16, voff, 0, 0, 0);
            }
        }
    }
    // ---- weights of one tap: piece (chunk cc, 16-row group g) = rows 16 g .. + 15 x 64 B; lane l moves slot l & 3 of row l >> 2, which
    // holds channel quad (l & 3) ^ ((row >> 2) & 3) of the chunk (conflict-free ds_read_b128, one 64-B request per row)
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    unsigned wvoff[W_PIECES];
    int wdst[W_PIECES];
#pragma unroll
    for (int h = 0; h < W_PIECES; ++h) {
        const int id = wave * W_PIECES + h, cc = id >> 2, g = id & 3;
        const int row = g * 16 + (lane >> 2), quad = (lane & 3) ^ ((lane >> 4) & 3);
        wvoff[h] = (unsigned)((((n0 + row) * TAPS) * CIN + cc * 16 + quad * 4) * 4);
        wdst[h] = __builtin_amdgcn_readfirstlane(W_BASE + cc * 4096 + g * 1024);
    }
    auto issue_weights = [&](int tap, int stage) {
#pragma unroll
        for (int h = 0; h < W_PIECES; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(smem + wdst[h] + stage * W_STAGE), 16, wvoff[h], tap * CIN * 4, 0, 0);
    };

    // ---- fragment addressing: lane (fi, fq) holds A[pixel 16 t + fi][k = 4 fq .. + 3], B[channel 16 wn + fi][same k] of a 16-wide K-step
    const int fi = lane & 15, fq = lane >> 4;
    int a_addr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = (wm * TM + i) * 16 + fi;
        m = m < BMR ? m : BMR - 1;                               // rows past the block: any valid pixel (masked in the epilogue)
        const int r = m / HW, c = m - r * HW;
        a_addr[i] = lds0 + (r * PW + c) * PPITCH + fq * 16;
    }
    const int b_addr = lds0 + W_BASE + (wn * 16 + fi) * 64 + ((fq ^ ((fi >> 2) & 3)) << 4);
    floatx4 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i][0] = floatx4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < LEAD; ++t) issue_weights(t, t);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if constexpr (XBN) {
        // the patch holds the producing convolution's pre-BatchNorm output: one pass over it in LDS turns the in-image pixels into the
        // activation (the zero border is the ACTIVATION's padding and stays).  130 pixels x 16 channel quads over 512 threads; a lane
        // keeps its quad (512 % 16 == 0), so its eight coefficients are loaded once.
        const int slot = tid & 15;
        floatx4 sc, sh;
        inbn_coeff4(p.in, slot * 4, sc, sh);
        for (int pix = tid >> 4; pix < PR * PW; pix += NW * 4) {
            const int prow = pix / PW, pcol = pix - prow * PW;
            const int iy = y0 - 1 + prow, ix = pcol - 1;
            if ((unsigned)iy < (unsigned)HW && (unsigned)ix < (unsigned)HW) {
                floatx4* q = reinterpret_cast<floatx4*>(smem + pix * PPITCH + slot * 16);
                *q = inbn_apply(*q, sc, sh);
            }
        }
        __syncthreads();
        inbn_commit(p.in);                                   // (block 0: mean / invstd / scale / shift for backward, running statistics)
    }

    // K-step s = (tap, 16-channel chunk), 36 of them, unrolled.  TMW = this wave's real tiles (3 / 2).
    auto run = [&](auto TMW_C) {
        constexpr int TMW = decltype(TMW_C)::value;
        floatx4 af[2][TM], bf[2];
        auto read_frags = [&](auto S) {
            constexpr int s = decltype(S)::value, tap = s / NCC, cc = s % NCC, set = s & 1;
            constexpr int aoff = ((tap / 3) * PW + tap % 3) * PPITCH + cc * 64;
            bf[set] = lds_read16<(tap % W_STAGES) * W_STAGE + cc * 4096>(b_addr);
            static_for<TMW>([&](auto I) { af[set][decltype(I)::value] = lds_read16<aoff>(a_addr[decltype(I)::value]); });
        };
        read_frags(std::integral_constant<int, 0>{});
        static_for<TAPS * NCC>([&](auto S) {
            constexpr int s = decltype(S)::value, tap = s / NCC, cc = s % NCC, set = s & 1;
            if constexpr (cc == 0 && tap + LEAD < TAPS) issue_weights(tap + LEAD, (tap + LEAD) % W_STAGES);   // its stage was read during tap - 1
            if constexpr (cc == NCC - 1 && tap + 1 < TAPS) {
                // tap boundary: my reads of this tap are complete, my pieces of the next tap have landed (the two of tap + 2 may be in
                // flight) -- and behind the barrier everybody's
                wait_lgkmcnt<0>();
                wait_vmcnt<(LEAD > 1 && tap + LEAD < TAPS) ? W_PIECES * (LEAD - 1) : 0>();
                __builtin_amdgcn_s_barrier();
                read_frags(std::integral_constant<int, s + 1>{});
            } else if constexpr (s + 1 < TAPS * NCC) {
                read_frags(std::integral_constant<int, s + 1>{});
                wait_lgkmcnt<TMW + 1>();                          // the fragments of step s (issued one step ago) are in
            } else {
                wait_lgkmcnt<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TMW; ++i)
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[set][i][e], bf[set][e], acc[i][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    if (wm == 0) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 2>{});
    __syncthreads();

    const int mlim = m0 + BMR < p.M ? m0 + BMR : p.M;
    igemm_epilogue<BM, BN, TM, 1, 2, NW>(p.epi, acc, m0, n0, mlim, p.Cout, smem);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_conv_img_f32(const float* x, const float* w, float* y, const ConvGeom& g, const ConvEpilogue& e, hipStream_t stream, const InBn& in) {
    if (g.R != 3 || g.S != 3 || g.stride != 1 || g.pad != 1 || g.Hin != HW || g.Win != HW || g.Hout != HW || g.Wout != HW) return 0;
    if (g.Cin != CIN || g.Cout % BN != 0) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_IMG_F32", 1);    // 0 = off (ablation build)
    if (mode == 0) return 0;
    const double xb = 4.0 * g.B * HW * HW * CIN, wb = 4.0 * g.Cout * TAPS * CIN;
    if (xb >= 2147483000.0 || wb >= 2147483000.0) return 0;
    ImgF32Args p;
    p.x = x; p.w = w; p.epi = make_epi(y, e);
    p.M = g.M(); p.Cout = g.Cout; p.tilesN = g.Cout / BN;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.in = in;
    const unsigned blocks = (unsigned)(g.B * (HW / ROWS) * p.tilesN);
    if (blocks > 512) return 0;                                  // more rounds: the implicit-GEMM tiles win (see the header)
    note_launch("conv_img_f32");
    prof_launch_begin(2, 2.0 * p.M * p.Cout * TAPS * CIN, 4.0 * ((double)p.M * CIN + (double)p.Cout * TAPS * CIN + (double)p.M * p.Cout), stream);
    if (in.on()) hipLaunchKernelGGL(conv_img_f32_kernel<true>, dim3(blocks), dim3(NW * 64), 0, stream, p);
    else hipLaunchKernelGGL(conv_img_f32_kernel<false>, dim3(blocks), dim3(NW * 64), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
