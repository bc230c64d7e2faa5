// 3x3 convolution of the 64-input-channel layers on the bf16 matrix cores with BOTH operands resident in LDS: one block = half a
// 24 x 24 feature map (12 rows, 288 output pixels) x 64 output channels; the 14 x 26 halo patch (all 64 channels) and all nine taps of
// the 64 x 64 weights are DMA-ed once, then the 18 K-steps (tap, 32-channel half) run without a barrier or a global access.
//
// Reference operators: the 3x3 / stride 1 / pad 1 nn.Conv2d forwards of layer1's BasicBlocks and of layer2's first convolution
// (resnet.py:31-47, 64 input channels, 24 x 24 maps) and the dgrads (train.py:132) that contract over 64 channels, in plain-bf16 plans
// at the batch sizes of BASELINE configs[2] / [4] (128 transitions per GPU).
//
// Why: N = 64 and K = 576 make these the least intense matrix problems of the network (5.4 GFLOP at B = 128); the LDS-DMA
// implicit-GEMM tile (144 x 64, conv_igemm_bf16_dma.hip) re-stages every input pixel once per tap and the weights once per 144 rows:
// 120 MB through L2 -> LDS per launch, 37 us = 0.15 PF/s.  Here a launch stages 2 B blocks x (52 KB patch + 83 KB weights) -- the
// patch once instead of nine times.
// LDS: patch [14 rows][26 columns][128 B + 16 B pad], weights [tap][64 rows][128 B + 16 B pad]: with 144-byte rows the 16 lanes a
// ds_read_b128 serves together (16 consecutive pixels / output channels, same 16-byte k slot) fall into 16 different 4-bank groups;
// the 16-pixel tiles that wrap an image row lose two lanes to conflicts.  A DMA piece is 1 KB of that padded array: every lane
// works out which (row, 16-byte slot) its 16 bytes are -- the pad slots and the patch border carry an out-of-range offset (zero fill).
// 4 waves = 2 (pixel halves: nine 16-pixel tiles) x 2 (32 output channels): 18 accumulator tiles, 11 fragment reads per 18 MFMAs.
// Measured (tools/c64_check.py, B = 128, 64 -> 64, per launch through the profiler's event pairs): 19.1 us against 26.1 (144 x 64 LDS-DMA
// tile); 64 -> 128: 31.6 | 42.4.  Where the 19 us go (a throw-away build with runtime switches): an empty kernel with this LDS footprint
// 6.7; + staging alone 0.8, + K loop alone 2.0, + epilogue alone 4.3; staging + K loop together 5.9 (12.6 - 6.7) -- one block per CU:
// nothing overlaps the three phases, and the 64-byte row segments of the 144 x 32 wave tiles make the epilogue the longest of them.
// Starting every block's weight staging at a different piece (L2 channel camping on the shared 83 KB) changed nothing.
// Arithmetic: v_mfma_f32_16x16x32_bf16 on the same bf16 operands as the other plain-bf16 kernels, fp32 accumulation, K order (tap,
// channel); epilogue igemm_epilogue.h (staged form).
#include <cstdlib>

#include "common.h"
#include "igemm_bf16_args.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int HW = 24, PW = 26, ROWS = 12, PR = ROWS + 2;
constexpr int CIN = 64, BN = 64, NW = 4, WM = 2, TAPS = 9;
constexpr int BM = ROWS * HW;                            // 288
constexpr int TM = BM / WM / 16, TN = BN / (NW / WM) / 16;   // 9 x 2 MFMA tiles per wave
constexpr int PITCH = CIN * 2 + 16;                      // 144 B per pixel / weight row
constexpr int PATCH_BYTES = PR * PW * PITCH;             // 52 416
constexpr int PATCH_PIECES = (PATCH_BYTES + 1023) / 1024;   // 52
constexpr int W_BASE = PATCH_PIECES * 1024;              // 53 248
constexpr int W_BYTES = TAPS * BN * PITCH;               // 82 944
constexpr int W_PIECES = W_BYTES / 1024;                 // 81
constexpr int SMEM_LOOP = W_BASE + W_PIECES * 1024;      // 136 192
constexpr int SMEM_EPI = staged_epilogue_smem<BN, TN, WM, NW, 3>();
constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
static_assert(W_PIECES * 1024 == W_BYTES && SMEM <= 160 * 1024, "LDS layout");

__global__ void __launch_bounds__(NW * 64) igemm_bf16_c64_kernel(const IgemmBfArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x;
    const int hb = tile / p.tilesN, tn = tile - hb * p.tilesN;       // half-image index, 64-channel tile
    const int img = hb >> 1, y0 = (hb & 1) * ROWS;
    const int m0 = hb * BM, n0 = tn * BN;

    // ---- stage both operands (1-KB pieces round-robin over the waves; nothing is read before all of it has landed)
    {
        __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x[0]), 0, p.x_bytes, 0x00020000);
        for (int q = wave; q < PATCH_PIECES; q += NW) {
            const int off = q * 1024 + lane * 16;
            const int pix = off / PITCH, slot = (off - pix * PITCH) >> 4;        // slot 8 = the pad
            const int prow = pix / PW, pcol = pix - prow * PW;
            const int iy = y0 - 1 + prow, ix = pcol - 1;
            const bool ok = pix < PR * PW && slot < 8 && (unsigned)iy < (unsigned)HW && (unsigned)ix < (unsigned)HW;
            const unsigned voff = ok ? (unsigned)((((img * HW + iy) * HW + ix) * CIN + slot * 8) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(smem + q * 1024), 16, voff, 0, 0, 0);
        }
        __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w[0]), 0, p.w_bytes, 0x00020000);
        for (int q = wave; q < W_PIECES; q += NW) {
            const int off = q * 1024 + lane * 16;
            const int row = off / PITCH, slot = (off - row * PITCH) >> 4;        // row = tap * 64 + output channel
            const int tap = row >> 6, co = row & 63;
            const unsigned voff = slot < 8 ? (unsigned)((((n0 + co) * TAPS + tap) * CIN + slot * 8) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(smem + W_BASE + q * 1024), 16, voff, 0, 0, 0);
        }
    }

    // ---- fragment addressing: lane (fi, fq) holds A[pixel 16 t + fi][k = 8 fq .. + 7], B[channel fi][same k] of a 32-wide K-step
    const int fi = lane & 15, fq = lane >> 4;
    int a_off[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wm * TM + i) * 16 + fi;
        const int r = m / HW, c = m - r * HW;
        a_off[i] = (r * PW + c) * PITCH + fq * 16;
    }
    const int b_off = W_BASE + (wn * (TN * 16) + fi) * PITCH + fq * 16;          // + j * 16 * PITCH + tap * 64 * PITCH + half * 64
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        const int aoff = ((tap / 3) * PW + tap % 3) * PITCH;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(smem + b_off + (tap * BN + j * 16) * PITCH + half * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + a_off[i] + aoff + half * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    igemm_epilogue_staged<BM, BN, TM, TN, WM, NW, 3, true>(p.epi, acc, m0, n0, p.M, p.Cout, smem);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_conv_igemm_bf16_c64(const IgemmBfArgs& a, hipStream_t stream) {
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.Hin != HW || a.Win != HW || a.Hout != HW || a.Wout != HW) return 0;
    if (a.Cin != CIN || a.Cout % BN != 0 || a.M % BM != 0 || a.x_bytes >= 0x7FFF0000u) return 0;
    static const int mode = SIMQ_TUNE_INT("SIMQ_BF16_C64", 1);   // 0 = off (ablation build)
    if (mode == 0) return 0;
    int fbm = 0, fbn = 0;
    const bool forced = bf16_forced_tile(a, &fbm, &fbn);
    if (forced && !(fbm == BM && fbn == BN)) return 0;
    const long blocks = (long)(a.M / BM) * (a.Cout / BN);
    if (!forced && blocks < 200) return 0;                      // one block per CU: needs (nearly) all of them
    // 64 -> 128 (layer2's first convolution): the half-map image tile (conv_igemm_bf16_img.hip, 128 output channels per block: the weights
    // are staged once per 12 image rows instead of twice) measured 35.3 us against 42.0 here at B = 128 (tools/img_half_check.py)
    if (!forced && a.Cout % 128 == 0 && (a.M / (HW * HW)) * 2 * (a.Cout / 128) >= 200 && (a.M / (HW * HW)) * (a.Cout / 128) < 200) return 0;
    IgemmBfArgs p = a;
    p.tilesN = p.Cout / BN;
    p.xcd_chunk = 0;
    note_launch("igemm_bf16_c64");
    prof_launch_begin(2, 2.0 * p.M * p.Cout * p.K,
                      4.0 * ((double)p.M / (p.Hout * p.Wout) * p.Hin * p.Win * p.Cin + (double)p.Cout * p.K + (double)p.M * p.Cout), stream);
    hipLaunchKernelGGL(igemm_bf16_c64_kernel, dim3((unsigned)blocks), dim3(NW * 64), 0, stream, p);
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
