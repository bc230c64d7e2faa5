// Batched fp32 GEMM on the bf16 matrix cores of gfx950 through an EXACT three-way operand split ("bf16x6").
//
//   y_g[M][N] = x_g[M][K] * w_g[N][K]^T        g = 0 .. P-1   (row-major fp32, K contiguous: the transform-domain contractions of the
//                                                               Winograd layers, conv_winograd.hip -- forward, dgrad and weight gradient)
//
// Why: the fp32 step's dominant kernel is igemm_conv_kernel<64,64,.,batched> on v_mfma_f32_16x16x4_f32 -- 64 flop / cycle / SIMD, and the
// kernel already runs at 0.83 of that rate on its marginal flop (profiles/r05_gemm_batched_probe.txt).  The bf16 matrix pipe of the same
// CU is 16 x faster.  An fp32 value has 24 significant bits, a bf16 value 8, and the two formats share their exponent range, so
//       v = v0 + v1 + v2,   v0 = trunc_bf16(v),  v1 = trunc_bf16(v - v0),  v2 = v - v0 - v1      (every subtraction exact, v2 a bf16 value)
// holds EXACTLY for every finite fp32 v (down to 2^-109: below it the last piece underflows to zero, an error below 2^-133), and every
// product of two pieces is exact in fp32 (8 x 8 significant bits).  a*b = sum_ij a_i*b_j; the three terms with i + j >= 3 are below
// 2^-24 |a*b| -- half an fp32 ulp of the product, i.e. below what ONE rounding of the fp32 FMA chain loses -- so six bf16 MFMAs
//       a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0),         smallest terms first, all into the fp32 accumulator of the matrix core,
// reproduce the fp32 contraction to fp32 round-off (measured against fp64 beside the fp32-MFMA kernel: tests/test_gpu_ops.py,
// tools/gemm_batched_probe.py) at 6/16 of its matrix-pipe time.  Nothing is stored in reduced precision: operands are read as fp32 and
// split while they are staged into LDS, the accumulators and the output are fp32.
//
// Block: 256 threads = 2 x 2 wave64, block tile 128 x 128, wave tile 64 x 64 = 2 x 2 tiles of v_mfma_f32_32x32x16_bf16; K-step 16.
// Loader: 4 threads per row x float4 (64 B of a row per K-step, as conv_igemm.hip), two K-steps in flight in registers; a thread splits
// its float4 into three 8-byte groups (and / sub / perm: 5.5 VALU per element, hidden beside the other wave's MFMAs -- two blocks per CU)
// and writes them with ds_write_b64.  LDS per stage and operand plane: [k half h][row ^ 8h][16 B] -- lane (row = l & 31, h = l >> 5) of a
// fragment load finds its 8 k-values in ONE 16-byte slot, every 16-lane group of the ds_read_b128 hits 16 distinct slots, and the XOR
// keeps the two k halves of the ds_write_b64 groups on different banks.  2 stages x 24 KB; one barrier per K-step.
#include "common.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3Args {
    const float* x;
    const float* w;
    float* y;
    int M, N, K;
    long gx, gw, gy;          // elements between consecutive problems
    unsigned x_bytes, w_bytes;
    int tilesN;
    int plane_xcd;
};

// the three bf16 pieces of four consecutive values, as packed pairs: out[p][0] = (v.x, v.y) of piece p, out[p][1] = (v.z, v.w)
__device__ __forceinline__ void split3(const float4 v, unsigned (&out)[3][2]) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned pc[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned b0 = __float_as_uint(f[e]) & 0xffff0000u;
        const float r1 = f[e] - __uint_as_float(b0);                 // exact: the low 16 significant bits
        const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(b1);                   // exact, at most 8 significant bits: a bf16 value
        pc[0][e] = b0; pc[1][e] = b1; pc[2][e] = __float_as_uint(r2);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        // bytes 2,3 of the first value -> low half, bytes 2,3 of the second -> high half
        out[p][0] = __builtin_amdgcn_perm(pc[p][1], pc[p][0], 0x07060302u);
        out[p][1] = __builtin_amdgcn_perm(pc[p][3], pc[p][2], 0x07060302u);
    }
}

// WM x WN waves of 64 x 64 each: block tile (64 WM) x (64 WN), 64 WM WN threads; OCC = waves per SIMD the register budget is held to.
//   <2, 2, 2>  128 x 128, 256 threads, two blocks per CU     <2, 4, 1>  128 x 256, 512 threads, one block per CU: the same eight waves per CU and
//   the same tile rounds per launch, a quarter fewer operand elements staged (and split) per MFMA
template <int WM, int WN, int OCC, bool ABL_NOSPLIT = false>
__global__ void __launch_bounds__(64 * WM * WN, OCC) gemm_split3_kernel(const Split3Args p) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    constexpr int TM = 2, TN = 2;                                  // 32x32 MFMA tiles per wave and dimension
    constexpr int A_PLANE = 2 * BM * 16, B_PLANE = 2 * BN * 16;    // bytes: [2 k-halves][rows][16 B]
    constexpr int STAGE = 3 * (A_PLANE + B_PLANE);
    constexpr int RPP = NT / 4;                                    // loader: 4 threads per row, RPP rows per pass
    static_assert(BM % RPP == 0 && BN % RPP == 0, "the loader passes must tile the block rows");
    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tile = blockIdx.x, plane = blockIdx.y;
    if (p.plane_xcd) {
        // whole planes per XCD (as igemm_conv_kernel's batched form): workgroups are dealt round-robin to the 8 XCDs in linear order; the
        // j-th block an XCD receives works on plane k + 8 (j / tiles), so that one plane's operands live in ONE L2; the planes left over
        // when their count is not a multiple of 8 are shared by 8 / rem XCDs each, a contiguous range of tiles per XCD
        const unsigned gx = gridDim.x, P = gridDim.y;
        const unsigned lin = blockIdx.y * gx + blockIdx.x, k = lin & 7u, j = lin >> 3;
        const unsigned full = P >> 3, rem = P & 7u;
        if (j < full * gx) {
            plane = (int)(k + 8u * (j / gx));
            tile = (int)(j % gx);
        } else {
            const unsigned share = 8u / rem, jj = j - full * gx;
            plane = (int)(8u * full + (k % rem));
            tile = (int)((k / rem) * (gx / share) + jj);
        }
    }
    const int tile_m = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / 16;

    const float* px = p.x + plane * p.gx;
    const float* pw = p.w + plane * p.gw;
    // bounds-checked buffer loads: rows past M lie past the end of the plane and read as zeros
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pw), 0, p.w_bytes, 0x00020000);

    const int lrow = tid >> 2, kq = tid & 3;
    unsigned aoff[A_PASSES], boff[B_PASSES];
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) aoff[ps] = (unsigned)(((m0 + lrow + RPP * ps) * p.K + kq * 4) * 4);
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) boff[ps] = (unsigned)(((n0 + lrow + RPP * ps) * p.K + kq * 4) * 4);
    // where this thread's 8-byte groups go inside a plane: k half h = kq >> 1, first / second 8 bytes of the slot = kq & 1
    const int wh = kq >> 1;
    unsigned awr[A_PASSES], bwr[B_PASSES];
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) awr[ps] = (unsigned)(wh * (BM * 16) + (((lrow + RPP * ps) ^ (wh << 3)) << 4) + (kq & 1) * 8);
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) bwr[ps] = (unsigned)(3 * A_PLANE + wh * (BN * 16) + (((lrow + RPP * ps) ^ (wh << 3)) << 4) + (kq & 1) * 8);

    float4 va[2][A_PASSES], vb[2][B_PASSES];
    auto load_tile = [&](auto set_c, int kt) {
        constexpr int SET = decltype(set_c)::value;
        const unsigned koff = (unsigned)kt * 64u;                  // 16 floats per K-step; past the last step: past the row, but inside the plane
        const bool live = kt < nk;                                 // ... so dead steps are masked to the out-of-range offset explicitly
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps)
            va[SET][ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, live ? aoff[ps] + koff : 0xFFFFFFFFu, 0, 0));
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps)
            vb[SET][ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, live ? boff[ps] + koff : 0xFFFFFFFFu, 0, 0));
    };
    auto store_tile = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        char* st = smem + buf * STAGE;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            unsigned pk[3][2];
            if constexpr (ABL_NOSPLIT) {      // timing ablation (wrong results): what the kernel would cost with operands that arrive split
                const float4 v = va[SET][ps];
                pk[0][0] = __float_as_uint(v.x); pk[0][1] = __float_as_uint(v.y); pk[1][0] = __float_as_uint(v.z); pk[1][1] = __float_as_uint(v.w);
                pk[2][0] = pk[0][0]; pk[2][1] = pk[1][1];
            } else
            split3(va[SET][ps], pk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(st + pl * A_PLANE + awr[ps]) = make_uint2(pk[pl][0], pk[pl][1]);
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            unsigned pk[3][2];
            if constexpr (ABL_NOSPLIT) {
                const float4 v = vb[SET][ps];
                pk[0][0] = __float_as_uint(v.x); pk[0][1] = __float_as_uint(v.y); pk[1][0] = __float_as_uint(v.z); pk[1][1] = __float_as_uint(v.w);
                pk[2][0] = pk[0][0]; pk[2][1] = pk[1][1];
            } else
            split3(vb[SET][ps], pk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(st + pl * B_PLANE + bwr[ps]) = make_uint2(pk[pl][0], pk[pl][1]);
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane (row = l & 31, h = l >> 5) reads k = 8h .. 8h+7 of its row from one 16-byte slot
    const int fr = lane & 31, fh = lane >> 5;
    unsigned ard[TM], brd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) ard[i] = (unsigned)(fh * (BM * 16) + (((wm * 64 + i * 32 + fr) ^ (fh << 3)) << 4));
#pragma unroll
    for (int j = 0; j < TN; ++j) brd[j] = (unsigned)(3 * A_PLANE + fh * (BN * 16) + (((wn * 64 + j * 32 + fr) ^ (fh << 3)) << 4));

    bf16x8 af[3][TM], bf[3][TN];
    auto read_frags = [&](int buf) {
        const char* st = smem + buf * STAGE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(st + pl * A_PLANE + ard[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(st + pl * B_PLANE + brd[j]);
        }
    };
    auto mfma_step = [&]() {
        // smallest products first; consecutive MFMAs go to different accumulators (no dependent-issue stall)
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]][i], bf[PB[t]][j], acc[i][j], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // pipeline: K-step k's fragments come from LDS stage k & 1; tile k+1 is split and stored into the other stage while tile k+2 / k+3 are
    // in flight from L2 / HBM in the two register sets.  Loads and stores are unconditional (zeros past the end): counted vmcnt waits.
    load_tile(S0{}, 0);
    load_tile(S1{}, 1);
    store_tile(S0{}, 0);
    load_tile(S0{}, 2);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        read_frags(0);
        store_tile(S1{}, 1);                          // tile k+1 -> stage 1 (every wave left stage 1 before the last barrier)
        load_tile(S1{}, kt + 3);
        mfma_step();
        __syncthreads();
        read_frags(1);
        store_tile(S0{}, 0);                          // tile k+2 -> stage 0
        load_tile(S0{}, kt + 4);
        mfma_step();
        __syncthreads();
    }
    if (kt < nk) {
        read_frags(0);
        mfma_step();
    }

    // D[row][col]: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a register is 32 consecutive floats of two rows
    float* py = p.y + plane * p.gy;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (row < p.M) {
                float* dst = py + (size_t)row * p.N + n0 + wn * 64 + fr;
#pragma unroll
                for (int j = 0; j < TN; ++j) dst[j * 32] = acc[i][j][r];
            }
        }
}

}  // namespace

// whether the split form is taken for this problem (launch_gemm_batched falls back to the fp32-MFMA kernel otherwise)
bool gemm_split3_eligible(int M, int N, int K, int batch) {
    return K % 16 == 0 && N % 128 == 0 && M >= 64 && batch >= 1;
}

int launch_gemm_batched_split3(const float* x, const float* w, float* y, int M, int N, int K, int batch, hipStream_t stream, const LaunchTune& tune) {
    SIMQ_REQUIRE(gemm_split3_eligible(M, N, K, batch), "gemm_split3: M=%d N=%d K=%d batch=%d not supported", M, N, K, batch);
    Split3Args a;
    a.x = x; a.w = w; a.y = y; a.M = M; a.N = N; a.K = K;
    a.gx = (long)M * K; a.gw = (long)N * K; a.gy = (long)M * N;
    const double xb = 4.0 * M * K, wb = 4.0 * N * K;
    SIMQ_REQUIRE(xb < 4294967000.0 && wb < 4294967000.0 && 4.0 * (M + 128) * K < 4294967000.0, "gemm_split3: operand exceeds the 4 GiB buffer-addressing limit");
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    // tile: 128 x 256 (eight waves) where the columns allow; tune.force_bm / force_bn = 128 / 128 | 128 / 256 select a form for
    // tools/gemm_batched_probe.py and the per-kernel tests.  Measured and not kept: 128 x 128 held to three waves per SIMD (spills 21 registers,
    // 8-13 % slower); a 64 x 128 tile of two waves for the launches with few blocks (3-40 % slower on every shape of the step,
    // gpurun_out/split3_probe_small.log -> profiles/r06_gemm_split3_probe.txt)
    // (measured per shape, profiles/r06_gemm_split3_probe.txt: the wide tile wins where a block has >= 32 K-steps to amortise its prologue
    // and epilogue -- one block per CU, nothing beside it -- and the launch still has two rounds of blocks: M = 1152, N = K = 512 116.7 us
    // against 127.0; K = 256 or a single column of tiles: the two co-resident 128 x 128 blocks are 5-15 % faster)
    const long wide_blocks = (long)((M + 127) / 128) * (N / 256) * batch;
    const int bm = 128;
    int bn = (N % 256 == 0 && K >= 512 && wide_blocks >= 512) ? 256 : 128;
    if (tune.force_bm == 128 && (tune.force_bn == 128 || (tune.force_bn == 256 && N % 256 == 0))) bn = tune.force_bn;
    a.tilesN = N / bn;
    const int tilesM = (M + bm - 1) / bm;
    const int tiles = tilesM * a.tilesN;
    // whole planes per XCD need the tile count of the shared planes to divide (see the kernel); otherwise launch order
    const int rem = batch & 7;
    const bool rem_ok = rem == 0 || ((rem == 1 || rem == 2 || rem == 4) && tiles % (8 / rem) == 0);
    a.plane_xcd = (tune.plane_xcd && batch >= 8 && rem_ok && ((long)tiles * batch) % 8 == 0) ? 1 : 0;
    note_launch("gemm_split3_batched");
    prof_launch_begin(0, 2.0 * M * N * K * batch, 4.0 * batch * ((double)M * K + (double)N * K + (double)M * N), stream);
#ifdef SIMQ_ABLATIONS
    if (tune.force_bm == 128 && tune.force_bn == 130) { a.tilesN = N / 128; hipLaunchKernelGGL((gemm_split3_kernel<2, 2, 2, true>), dim3((unsigned)(tilesM * a.tilesN), (unsigned)batch), dim3(256), 0, stream, a); SIMQ_CHECK_LAUNCH(); return 0; }
#endif
    if (bn == 256) hipLaunchKernelGGL((gemm_split3_kernel<2, 4, 1>), dim3((unsigned)tiles, (unsigned)batch), dim3(512), 0, stream, a);
    else {
        // (ablation build: SIMQ_SPLIT3_PAD_LDS = bytes of unused dynamic LDS per block -- 40000 leaves room for ONE block per CU, so that a
        // co-running HBM-bound kernel gets two thirds of the register file: A/B of "GEMM alone faster" against "the pair faster")
        static const int pad = SIMQ_TUNE_INT("SIMQ_SPLIT3_PAD_LDS", 0);
        hipLaunchKernelGGL((gemm_split3_kernel<2, 2, 2>), dim3((unsigned)tiles, (unsigned)batch), dim3(256), (size_t)pad, stream, a);
    }
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
