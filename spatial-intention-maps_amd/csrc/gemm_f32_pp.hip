// Batched fp32 GEMM on the matrix cores of gfx950, "ping-pong" form:  y_g[M][N] = sum_k x_g[M][k] * w_g[N][k]  (both operands
// K-contiguous), g = blockIdx.y.  These are the transform-domain contractions of the Winograd layers (conv_winograd.hip: 16 or 36
// GEMMs per 3x3 convolution of reference resnet.py:31-47 -- forward, dgrad and, with the roles of the operands swapped, wgrad),
// the dominant kernel of the fp32 step.
//
// The register-staged 64 x 64 form of conv_igemm.hip (igemm_conv_kernel<64,64,true,true>) sits at 0.69 of the fp32 matrix peak on
// these short-K problems (K = Cin = 128 .. 512): every wave interleaves global loads, LDS stores, fragment reads and MFMAs, and
// a block-wide barrier per K-step keeps the four of them in phase.  Here, as in conv_igemm_bf16_pp.hip:
//   * operands go HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane; rows past M are zero-filled by the bounds check):
//     no staging registers, no LDS stores, counted vmcnt;
//   * 8 waves; the two waves that share a SIMD (w and w + 4 = upper / lower half of the tile rows) run ONE BARRIER APART: while
//     one is in its MFMA segment (32 x v_mfma_f32_16x16x4_f32 = 1024 matrix-pipe cycles, s_setprio 1, no memory instruction) the
//     other reads its 6 fragments (ds_read_b128) and issues its 2 DMA pieces of the K-tile three ahead;
//   * tile 128 x 128, K-tile 16 (one 64-B LDS row per operand row, 16-B slots XOR-swizzled by row bits: conflict-free
//     ds_read_b128), FOUR stages of 16 KB = 64 KB, <= 128 registers: TWO blocks per CU, so a block's prologue / epilogue is covered
//     by its neighbour;
//   * v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain: results are bit-identical to the register-staged kernel's whenever the
//     K order per accumulator is the same (it is: k = 4q + e, e inner), so the fp32 parity bars are untouched.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace simq {

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, NW = 8, WM = 2, WN = 4;
constexpr int WTM = BM / WM, WTN = BN / WN;             // 64 x 32
constexpr int TM = WTM / 16, TN = WTN / 16;             // 4 x 2 MFMA tiles per wave
constexpr int BK = 16;                                  // K-tile: 64-B rows
constexpr int NBUF = 4;
constexpr int GROUPS = (BM + BN) / 16;                  // 16 pieces of 16 rows x 64 B per stage
constexpr int NI = GROUPS / NW;                         // 2 per wave: piece `wave` of A, piece `wave` of B
constexpr int STAGE = (BM + BN) * 64;                   // 16 KB
constexpr int SMEM = NBUF * STAGE;                      // 64 KB

static_assert(GROUPS % NW == 0 && NI == 2 && BM / 16 == NW, "one A piece and one B piece per wave and K-tile");

struct GemmPpArgs {
    const float* x; const float* w; float* y;
    int M, N, K;
    long gx, gw, gy;                                    // elements between consecutive problems
    unsigned x_bytes, w_bytes, y_bytes;                 // per problem (bounds-checked buffer addressing)
    int tilesN, xcd_chunk;
    int tiles, batch, full;                             // tiles per problem; problems; problems scheduled one-per-XCD (multiple of 8)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// physical 16-B slot of logical k-chunk q in row r: q ^ {0, 2, 3, 1}[(r >> 2) & 3]
__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

template <int DBG = 0>
__global__ void __launch_bounds__(NW * 64, 2) gemm_f32_pp_kernel(const GemmPpArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;           // wm == wave group (waves w and w + 4 share a SIMD)
    // Block order (1-D grid; blocks go round-robin over the 8 XCDs, each with its own 4 MB L2).  A problem's w (N x K: 1 MB at
    // 512 x 512) is re-read by every row block and must stay L2-resident; x streams through once.  Problems 0 .. full-1 are
    // therefore scheduled ONE PER XCD: XCD k walks all tiles of problem 8r + k (N-tiles of a row block back to back), then of
    // 8(r+1) + k -- one w per L2 at a time instead of a slice of every problem in flight.  The remaining batch % 8 problems are
    // spread as contiguous tile runs per XCD.
    int tile = blockIdx.x, g;
    if (tile < p.full * p.tiles) {
        const int xcd = tile & 7, idx = tile >> 3;
        const int r = idx / p.tiles;
        g = r * 8 + xcd;
        tile = idx - r * p.tiles;
    } else {
        tile -= p.full * p.tiles;
        g = p.full + tile / p.tiles;
        tile = tile % p.tiles;
        if (tile < 8 * p.xcd_chunk) tile = (tile & 7) * p.xcd_chunk + (tile >> 3);
    }
    const int tile_m = tile / p.tilesN, tile_n = tile % p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + g * p.gx), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w + g * p.gw), 0, p.w_bytes, 0x00020000);

    // ---- stager: this wave moves rows 16 * wave .. + 15 of A and of B; lane = (row >> 0 within piece) * 4 + physical slot
    const int rs = wave * 16 + (lane >> 2);
    const int kq = (lane & 3) ^ swz4(rs);               // logical k-chunk that lands in this lane's physical slot
    const bool a_ok = m0 + rs < p.M;
    const unsigned abase = (unsigned)(((m0 + rs) * p.K + kq * 4) * 4);
    const unsigned bbase = (unsigned)(((n0 + rs) * p.K + kq * 4) * 4);
    char* const a_dst = smem + wave * 1024;
    char* const b_dst = smem + (BM / 16 + wave) * 1024;
    int kt_issue = 0;
    auto issue_tile = [&](int dbuf, bool live) {        // past the end (live = false) every lane is out of range: zeros, same vmcnt
        const unsigned koff = (unsigned)(kt_issue * BK * 4);
        const unsigned va = (live && a_ok) ? abase + koff : 0xFFFFFFFFu;
        const unsigned vb = live ? bbase + koff : 0xFFFFFFFFu;
        if constexpr (!(DBG & 1)) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(a_dst + dbuf * STAGE), 16, va, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(b_dst + dbuf * STAGE), 16, vb, 0, 0, 0);
        }
        ++kt_issue;
    };

    // ---- fragments: v_mfma_f32_16x16x4_f32 lane l holds A[i = l & 15][k = l >> 4]; a ds_read_b128 of slot fq gives k = 4 fq + e
    const int fi = lane & 15, fq = lane >> 4;
    const int ra = wm * WTM + fi, rb = wn * WTN + fi;
    const int a_off = ra * 64 + ((fq ^ swz4(ra)) << 4);
    const int b_off = BM * 64 + rb * 64 + ((fq ^ swz4(rb)) << 4);
    floatx4 af[TM], bf[TN];
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) issue_tile(t, t < nk);
    if constexpr (!(DBG & 1)) wait_vmcnt<(NBUF - 2) * NI>();
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0 from here on

    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + buf * STAGE;
        // ---------------- load segment ----------------
        if constexpr (!(DBG & 8)) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const floatx4*>(st + b_off + j * 1024);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const floatx4*>(st + a_off + i * 1024);
        }
        issue_tile((buf + NBUF - 1) & (NBUF - 1), kt + NBUF - 1 < nk);     // K-tile kt + 3 into the stage K-tile kt - 1 used
        if constexpr (!(DBG & 1)) wait_vmcnt<(NBUF - 2) * NI>();          // K-tile kt + 1 landed (this wave's pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        // ---------------- MFMA segment ----------------
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (DBG & 16) asm volatile("" :: "v"(af[i]), "v"(bf[j]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DBG & 2)) __builtin_amdgcn_s_barrier();
        buf = (buf + 1) & (NBUF - 1);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();          // group 0 waits for group 1's last MFMA segment
    wait_vmcnt<0>();
    __syncthreads();

    // ---- store: each wave transposes 32 rows of its tile at a time through a private LDS strip (the stages are idle), so the
    // output leaves as 16-byte pieces of 128-B row segments instead of 4-byte pieces of 64-B ones
    constexpr int LDW = WTN + 4;                        // strip row stride in floats (36: conflict-free b32 writes, aligned b128 reads)
    float* strip = reinterpret_cast<float*>(smem) + wave * (32 * LDW);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y + g * p.gy, 0, p.y_bytes, 0x00020000);
    const int cl = (lane & 7) * 4, rl = lane >> 3;      // 8 lanes x float4 per 32-float row, 8 rows per instruction
#pragma unroll
    for (int h = 0; h < TM / 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) strip[(ii * 16 + 4 * fq + r) * LDW + j * 16 + fi] = acc[h * 2 + ii][j][r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = k * 8 + rl;
            const int m = m0 + wm * WTM + h * 32 + row;
            const floatx4 v = *reinterpret_cast<const floatx4*>(strip + row * LDW + cl);
            const unsigned off = m < p.M ? (unsigned)((m * p.N + n0 + wn * WTN + cl) * 4) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, v), yr, off, 0, 0);
        }
    }
}

template <int DBG>
void launch(const GemmPpArgs& p, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL(gemm_f32_pp_kernel<DBG>, grid, dim3(NW * 64), 0, stream, p);
}

}  // namespace

// returns 1 when the launch was taken, 0 when the shape is not covered, < 0 on error
int try_gemm_batched_pp(const float* x, const float* w, float* y, int M, int N, int K, int batch, hipStream_t stream) {
    // OFF by default (SIMQ_F32_PP=1: F(2x2) layer4-sized problems, =2: every shape).  Measured against the register-staged 64 x 64
    // tile at B = 32 (tools/f32pp_check.py, kernel alone): 412 -> 371 us on layer4's 512 -> 512 F(2x2) GEMMs (94 -> 104 TF/s),
    // 204 -> 188 us at 256 -> 512, level on layer3 and on every F(4x4) problem (quarter-size problems quantise badly over
    // 2 x 256 block slots) -- and 0.0 % on the whole step (3098 vs 3100 tr/s), where the side-stream forward already fills the
    // matrix pipe's idle slots.  Ablations on the 512 -> 512 problem: no DMA 304 us, no fragment reads 312 us, no MFMA 132 us: the
    // matrix pipe alone runs at 127 TF/s inside this structure, the staging adds ~60 us that the ping-pong does not hide.
    static const int mode = SIMQ_TUNE_INT("SIMQ_F32_PP", 0);
    if (mode == 0 || N % BN != 0 || K % BK != 0 || K < BK) return 0;
    static const int min_m = SIMQ_TUNE_INT("SIMQ_F32_PP_MIN_M", 4096);
    static const int min_n = SIMQ_TUNE_INT("SIMQ_F32_PP_MIN_N", 512);
    if (mode != 2 && (M < min_m || N < min_n)) return 0;
    const double xb = 4.0 * M * K, wb = 4.0 * N * K, yb = 4.0 * M * N;
    if (xb >= 4294967000.0 || wb >= 4294967000.0 || yb >= 4294967000.0) return 0;
    GemmPpArgs p;
    p.x = x; p.w = w; p.y = y; p.M = M; p.N = N; p.K = K;
    p.gx = (long)M * K; p.gw = (long)N * K; p.gy = (long)M * N;
    p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; p.y_bytes = (unsigned)yb;
    p.tilesN = N / BN;
    const int tilesM = (M + BM - 1) / BM;
    const int tiles = tilesM * p.tilesN;
    // XCD-aware order (blocks go round-robin over the 8 XCDs, each with its own L2): one XCD walks a contiguous run of tiles, so the
    // N-tiles that share a row block read it through one L2
    static const bool remap = SIMQ_TUNE_INT("SIMQ_XCD_REMAP", 1) != 0;
    p.xcd_chunk = (remap && p.tilesN > 1 && tiles >= 64 && tiles % 8 == 0) ? tiles / 8 : 0;
    p.tiles = tiles; p.batch = batch;
    p.full = remap ? (batch / 8) * 8 : 0;
    note_launch("gemm_f32_pp");
    prof_launch_begin(0, 2.0 * M * N * K * batch, 4.0 * batch * ((double)M * K + (double)N * K + (double)M * N), stream);
    const dim3 grid((unsigned)tiles * (unsigned)batch);
#ifdef SIMQ_ABLATIONS      // timing ablations (tools/f32pp_check.py): compiled into libsimq_ablate.so only
    static const int dbg = SIMQ_TUNE_INT("SIMQ_F32_PP_DBG", 0);   // timing ablations
    switch (dbg) {
        case 1: launch<1>(p, grid, stream); break;       // no DMA
        case 2: launch<2>(p, grid, stream); break;       // no barriers
        case 8: launch<8>(p, grid, stream); break;       // no fragment reads
        case 16: launch<16>(p, grid, stream); break;     // no MFMAs
        default: launch<0>(p, grid, stream);
    }
#else
    launch<0>(p, grid, stream);
#endif
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
