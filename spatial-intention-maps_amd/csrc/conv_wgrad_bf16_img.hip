// Weight gradient of the 3x3 convolutions on the 24 x 24 maps, bf16 matrix cores of gfx950, IMAGE-TILE form: one block contracts
// ALL NINE filter taps of a 256 (co) x 32 (ci) tile from one staged copy of its operands.
//
//   dW[co][tap][ci] = sum_p dY[p][co] * X[p (+) tap][ci]        (wgrad half of loss.backward(), train.py:132)
//
// conv_wgrad_bf16_pp.hip gives every (tap, 256 x 256 tile) its own block: dY is staged nine times per ci-tile and X once per tap and
// co-tile -- 2.7 GB staged through L2 -> LDS for 151 MB of operands at layer4 / B = 128, 0.8 GB of it from the fabric (PMC: 837 MB per
// launch), MFMA-busy 0.42.  Here (the structure conv_igemm_bf16_img.hip proved for forward / dgrad):
//   * X: per image the 26 x 26 HALO PATCH of the block's 32 input channels is DMA-ed ONCE into LDS (two planes of 16 channels,
//     [plane][patch pixel][32 B]; out-of-image pixels zero-filled by the buffer range check); the nine taps read their B fragments
//     from it at shifted pixel positions -- an immediate offset in the ds_read, no re-staging, no per-tap address arithmetic.
//     Double-buffered: the patch of image i + 1 lands while image i is contracted.
//   * dY: streamed as stored, 32 pixels x 256 channels (16 KB) per step through a 4-stage ring (rows of 512 B, 32-B chunks
//     XOR-swizzled by the pixel row, swizzle folded into the DMA source offset), as in conv_wgrad_bf16_pp.hip.
//   * both contract over pixels, which is the strided index of both tensors: fragments come back through ds_read_b64_tr_b16
//     (transpose read: lane t of a 16-lane group addresses 4 channels of row t >> 2 and receives 4 consecutive rows of channel t).
//     The K order of an MFMA is free as long as A and B agree: lane group g holds pixels {4g .. 4g+3} u {16+4g .. 16+4g+3} of the step, so
//     that the two 16-lane groups a transpose read serves together sit 4 rows apart (rows of 32 B: banks 8 r mod 64 -- conflict-free
//     except where the two groups fall into different image rows, 1 pair in 6).
//   * 8 waves = 4 (co) x 2 (ci): wave tile 64 co x 16 ci x 9 taps = 36 accumulator tiles; per 32-pixel step 36 MFMAs
//     (v_mfma_f32_16x16x32_bf16) against 4 + 9 fragments = 26 transpose reads; the two waves of a SIMD run one barrier apart
//     (load segment | s_barrier | MFMA segment under s_setprio | s_barrier), counted vmcnt, no memory instruction beside the MFMAs.
//   * staged per step and block: 16 KB of dY + 2.4 KB of patch for 4.7 MFLOP (255 flop per staged byte; 131 for the per-tap tile);
//     dY is read Cin / 32 times, X Cout / 256 times (+ 17 % halo) -- and the Cin / 32 x Cout / 256 blocks that share an image range
//     are placed on ONE XCD (block b runs on XCD b mod 8), so those re-reads are L2 hits and HBM sees the operands about once.
// The pixel reduction is split over blocks by IMAGES: tiles x splits = 256 blocks, one per CU.  Every block leaves its 256 x 32 x 9
// partial tile in its own SLAB (plain 16-byte stores, [tap][co / 4][ci][4 co]: a lane's four accumulator rows are one float4, 16 lanes
// 256 contiguous bytes) and a second launch adds the slabs of a tile in a fixed order -- deterministic, and cheaper than the
// alternative it replaced: 75 MB of device-scope fp32 atomics, performed memory-side 64 bytes at a time, cost 40-55 us per launch
// (measured with the atomics switched off at run time: 315 -> 276 us at layer4, 126 -> 71 us at 256 -> 256 channels, B = 128).
// Without a slab buffer (op-level calls that pass none) the atomics remain.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace simq {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef short short8_ __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int TI = 256, TJ = 32, NW = 8;                 // block tile: 256 co x 32 ci (x 9 taps)
constexpr int BRB = 32;                                  // pixels per step (= MFMA k extent)
constexpr int NA = 4;                                    // wave tile: 4 x 16 co, 1 x 16 ci, 9 taps
constexpr int HW = 24, PW = 26;                          // map and halo-patch width
constexpr int STEPS = HW * HW / BRB;                     // 18 steps per image
constexpr int ROWB = 512;                                // bytes per staged dY pixel row (256 channels)
constexpr int STAGE = BRB * ROWB;                        // 16 KB per dY stage
constexpr int NBUF = 4;
constexpr int PPIECES = 22;                              // 1-KB DMA pieces per patch plane (676 patch pixels x 32 B, padded to 704)
constexpr int PLANE = PPIECES * 1024;
constexpr int PATCH = 2 * PLANE;                         // two 16-channel planes
constexpr int XSLOTS = (2 * PPIECES + NW - 1) / NW;      // patch pieces per wave and image (6)
constexpr int OFF_PATCH = NBUF * STAGE;                  // LDS map: dY ring | patch buffers 0, 1 | 1 KB for the idle DMA slots
constexpr int OFF_DUMMY = OFF_PATCH + 2 * PATCH;
constexpr int OFF_TABLE = OFF_DUMMY + 1024;              // u16 [512 threads][XSLOTS]: patch-DMA source offsets / 16 (0xFFFF: halo)
constexpr int SMEM = OFF_TABLE + XSLOTS * NW * 64 * 2;   // 162 816 B of the 163 840
constexpr int SLAB_FLOATS = 9 * TI * TJ;                 // one block's partial tile
constexpr int MAX_BLOCKS = 256;

struct WgradImgArgs {
    const uint16_t* x;
    const uint16_t* dy;
    float* dw;
    int Cin, Cout, K, B;
    int tilesI, tilesJ, splits, imgs_per_split;
    unsigned x_bytes, dy_bytes;
    float* slab;             // [blocks][9][64][32][4] partial tiles, or NULL: fp32 atomics into the zeroed dw
    int dbg;                 // run-time ablation bits (ablation build): 32 no atomics, 64 no main loop
};

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}) -- the body sees its index as a
// constant expression (immediate operands of inline assembly need one)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Transpose read through inline assembly (address = LDS byte address + immediate): seen as an ordinary LDS load, the compiler orders
// it behind EVERY LDS-DMA in flight (s_waitcnt vmcnt(0) in front of each step's reads -- it cannot know that the pieces in flight
// target other stages), which drains the prefetch pipeline once per step.  The ordering is the kernel's own: counted vmcnt + s_barrier
// before a stage is read, lgkmcnt(0) + sched_barrier before the fragments are used.
template <int IMM>
__device__ __forceinline__ short4_ tr_read(int addr) {
    static_assert(IMM >= 0 && IMM < 65536, "ds_read offset field");
    short4_ v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
    return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// DBG (timing ablations, libsimq_ablate.so only; results are wrong by construction): 1 no DMA in the loop, 8 no fragment reads, 16 no MFMAs
// (compile time); p.dbg (run time): 32 no stores / atomics, 64 no main loop
template <int DBG>
__global__ void __launch_bounds__(NW * 64, 2) wgrad_bf16_img_kernel(const WgradImgArgs p) {
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, wj = wave >> 2;             // co quarter; ci half == wave group (waves w and w + 4 share a SIMD)
    // block b = tile * splits + split: the blocks that share an image range (same split) have the same b mod 8 -> one XCD, one L2
    const int split = blockIdx.x % p.splits;
    int tile = blockIdx.x / p.splits;
    const int tj = tile % p.tilesJ;
    const int ti = tile / p.tilesJ;
    const int co0 = ti * TI, ci0 = tj * TJ;
    const int img0 = split * p.imgs_per_split;
    const int img1 = min(p.B, img0 + p.imgs_per_split);
    if (img0 >= img1) return;                            // block-uniform
    const int nimg = img1 - img0;

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, p.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dy), 0, p.dy_bytes, 0x00020000);

    // ---- dY stager: piece g = i * NW + wave (i = 0, 1) = stage rows 2g, 2g + 1; lane l moves physical 16-B slot (l & 31) of row
    // 2g + (l >> 5), i.e. logical 16-B unit ((slot >> 1) ^ (row & 7)) * 2 + (slot & 1) of the 512-B row
    unsigned vy[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * (i * NW + wave) + (lane >> 5);
        const int slot = lane & 31;
        const unsigned coff = (unsigned)(((((slot >> 1) ^ (row & 7)) << 1) | (slot & 1)) * 16);
        vy[i] = (unsigned)((img0 * HW * HW + row) * p.Cout + co0) * 2u + coff;
    }
    const unsigned step_y = (unsigned)(BRB * p.Cout * 2);
    auto issue_dy = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            char* dst = smem + stage * STAGE + (i * NW + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_void*)dst, 16, vy[i], 0, 0, 0);
            vy[i] += step_y;                             // (steps are issued in order; past the block's last image the rows are never read)
        }
    };

    // ---- patch stager: piece q = slot * NW + wave (slot 0 .. XSLOTS-1) = 32 patch pixels of plane q / PPIECES; lane l moves the 16-B
    // half (l & 1) of patch pixel (q % PPIECES) * 32 + (l >> 1).  Source offset relative to the image, 0xFFFFFFFF for halo / padding.
    // The per-lane offsets live in an LDS table (16-bit, in units of 16 B; they would cost six registers per lane that the
    // accumulators need): one ds_read_u16 per issued piece.
    // Table layout [thread][slot]: one base register per lane, the slot is an immediate offset of the read.
    uint16_t* table = reinterpret_cast<uint16_t*>(smem + OFF_TABLE) + tid * XSLOTS;
#pragma unroll
    for (int sl = 0; sl < XSLOTS; ++sl) {
        const int q = sl * NW + wave;
        const int plane = q / PPIECES;
        const int pp = (q - plane * PPIECES) * 32 + (lane >> 1);
        const int qy = pp / PW, qx = pp - qy * PW;
        const bool ok = q < 2 * PPIECES && qy >= 1 && qy <= HW && qx >= 1 && qx <= HW;
        table[sl] = ok ? (uint16_t)(((((qy - 1) * HW + (qx - 1)) * p.Cin + ci0 + plane * 16 + (lane & 1) * 8) * 2) >> 4) : (uint16_t)0xFFFF;
    }
    const unsigned img_x = (unsigned)(HW * HW * p.Cin * 2);
    // table entry of (this lane, slot s) -> buffer offset of the piece in image `img` (0xFFFFFFFF: zero-fill) and its DMA
    auto issue_patch_entry = [&](unsigned e, int s, int img, int buf) {    // img: image whose patch is fetched (block-uniform)
        const int q = s * NW + wave;
        const bool real = q < 2 * PPIECES && img < img1;
        char* dst = smem + (real ? OFF_PATCH + buf * PATCH + q * 1024 : OFF_DUMMY);
        const unsigned voff = (real && e != 0xFFFFu) ? (e << 4) + (unsigned)img * img_x : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)dst, 16, voff, 0, 0, 0);
    };
    auto issue_patch = [&](int s, int img, int buf) {    // prologue form: an ordinary (compiler-visible) table read
        issue_patch_entry(table[s], s, img, buf);
    };
    // ---- fragments.  Lane (t = lane & 15, g = lane >> 4); transpose read h (0, 1) covers step pixels 16 h + 4 g + (t >> 2) [rows] and
    // channels 4 (t & 3) .. + 3 of a 16-channel chunk; the lane receives 4 consecutive pixels of channel t.
    const int ft = lane & 15, fg = lane >> 4;
    // dY: byte offset inside a stage of chunk wi * 4 (a = 0), read h = 0.  Row r (h = 1: r + 16, same r & 7: + 16 * ROWB, an immediate);
    // chunk wi * 4 + a = (wi * 4) ^ a, so the swizzled position of chunk a is this offset XOR (a << 5) -- one register instead of eight
    const int a_r = 4 * fg + (ft >> 2);
    const int a_off0 = a_r * ROWB + (((wi * NA) ^ (a_r & 7)) << 5) + ((ft & 3) << 3);
    // X.  Image pixel (y, x) under tap (ky, kx) is patch pixel (y + ky, x + kx); the patch index of image pixel p is p + 2 (p / 24).  A
    // lane reads pixel p = c + 32 s at step s (c = 16 h + 4 g + (t >> 2) = 24 yc + xc), and p / 24 = yc + s + s / 3 + [xc >= 24 - 8 (s % 3)]:
    // everything but the bracket is the same for all lanes -- it goes into the IMMEDIATE offset of the read -- and the bracket is one of
    // three per-lane constants.  So: three base registers per read (s % 3 = 0, 1, 2), no address arithmetic in the loop.
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);     // LDS byte address of smem[0]
    // The slot-s table entry is read at the START of a load segment, in front of the 26 transpose reads (LDS returns in order: once at
    // most 15 LDS operations are outstanding the entry has arrived), so that the patch DMA does not queue behind the fragment traffic.
    // The address is re-derived from the lane id behind an opaque zero at every use: as a loop invariant it would cost a register.
    auto table_addr = [&]() -> int {
        int z;
        asm volatile("s_mov_b32 %0, 0" : "=s"(z));
        const int l_now = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
        return lds0 + OFF_TABLE + (wave * 64 + l_now) * (XSLOTS * 2);
    };
    int xb[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = 16 * h + 4 * fg + (ft >> 2);
        const int yc = c / HW, xc = c - yc * HW;
        const int b0 = lds0 + OFF_PATCH + wj * PLANE + (c + 2 * yc) * 32 + ((ft & 3) << 3);
        xb[h][0] = b0;
        xb[h][1] = b0 + (xc >= 16 ? 64 : 0);
        xb[h][2] = b0 + (xc >= 8 ? 64 : 0);
    }
    // dY: the four chunk positions of read h = 0 in stage 0 (stage and h are immediates)
    int ao[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) ao[a] = lds0 + (a_off0 ^ (a << 5));
    auto join = [&](short4_ lo, short4_ hi) -> bf16x8 {
        short8_ v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    bf16x8 af[NA], bf[9];
    floatx4 acc[9][NA];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[t][a] = floatx4{0.f, 0.f, 0.f, 0.f};

    // prologue: the first image's patch and dY steps 0 .. 2, landed and visible to everybody
    __syncthreads();                                     // (the offset table)
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) issue_patch(s, img0, 0);
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t) issue_dy(t);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (wj == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0 from here on

    // Two images per trip: 18 steps per image and a 4-stage ring make the dY stage (s + 2 (image & 1)) & 3 and the patch buffer image & 1
    // compile-time inside the unrolled body, so every LDS address below is a base register + an immediate.
    const int nloop = (p.dbg & 64) ? 0 : nimg;
    for (int im2 = 0; im2 < nloop; im2 += 2) {
        static_for<2>([&](auto HALF) {
            constexpr int half = decltype(HALF)::value;
            const int im = im2 + half;
            if (im < nloop) {                                            // (block-uniform)
                static_for<STEPS>([&](auto S) {
                    constexpr int s = decltype(S)::value;
                    constexpr int stage = (s + 2 * half) & (NBUF - 1);
                    constexpr int simm = (32 * s + 2 * (s + s / 3)) * 32;     // patch offset of the step (see xb)
                    const bool live = !(DBG & 8) || (im == 0 && s == 0);
                    // ---------------- load segment ----------------
                    constexpr bool patch_step = s >= 1 && s <= XSLOTS && !(DBG & 1);
                    unsigned entry = 0;
                    if constexpr (patch_step)
                        asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(entry) : "v"(table_addr()), "n"((s - 1) * 2) : "memory");
                    static_for<9>([&](auto T) {
                        constexpr int t = decltype(T)::value;
                        constexpr int imm = ((t / 3) * PW + (t % 3)) * 32 + simm;
                        if (live) bf[t] = join(tr_read<imm>(xb[0][s % 3]), tr_read<imm>(xb[1][s % 3]));
                    });
                    static_for<NA>([&](auto A) {
                        constexpr int a = decltype(A)::value;
                        if (live) af[a] = join(tr_read<stage * STAGE>(ao[a]), tr_read<stage * STAGE + 16 * ROWB>(ao[a]));
                    });
                    if (!(DBG & 1)) {
                        issue_dy((stage + NBUF - 1) & (NBUF - 1));      // dY of step + 3 into the stage step - 1 used
                        if constexpr (patch_step) {              // next image's patch, one piece per step
                            asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(entry) :: "memory");
                            issue_patch_entry(entry, s - 1, img0 + im + 1, (half + 1) & 1);
                        }
                    }
                    // counted wait: everything issued two steps ago or earlier has landed = at most the loads of this step and of the
                    // one before may be outstanding (2 dY pieces per step, + 1 patch piece in steps 1 .. XSLOTS)
                    constexpr int sp = (s + STEPS - 1) % STEPS;
                    constexpr int outstanding = 4 + ((s >= 1 && s <= XSLOTS) ? 1 : 0) + ((sp >= 1 && sp <= XSLOTS) ? 1 : 0);
                    wait_vmcnt<(DBG & 1) ? 0 : outstanding>();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    // ---------------- MFMA segment ----------------
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int a = 0; a < NA; ++a)
                            if (!(DBG & 16) || (im == 0 && s == 0)) acc[t][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[t], acc[t][a], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                });
                // the next image reads the other patch buffer
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int k = 0; k < 3; ++k) xb[h][k] += half == 0 ? PATCH : -PATCH;
            }
        });
    }
    if (wj == 0) __builtin_amdgcn_s_barrier();           // group 0 waits for group 1's last MFMA segment
    wait_vmcnt<0>();

    if (p.dbg & 32) {
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < NA; ++a) sum += acc[t][a][0] + acc[t][a][1] + acc[t][a][2] + acc[t][a][3];
        if (sum == 12345.678f) p.dw[0] = 1.f;
        return;
    }
    // C/D layout: col = lane & 15 (-> ci), row = 4 * (lane >> 4) + reg (-> co)
    if (p.slab) {
        float* base = p.slab + (size_t)blockIdx.x * SLAB_FLOATS + (size_t)(((wi * 16 + fg) * TJ + wj * 16 + ft) * 4);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int a = 0; a < NA; ++a)
                *reinterpret_cast<floatx4*>(base + (t * 64 + a * 4) * TJ * 4) = acc[t][a];
        return;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const size_t col = (size_t)t * p.Cin + ci0 + wj * 16 + ft;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = co0 + wi * (TI / 4) + a * 16 + 4 * fg + r;
                unsafeAtomicAdd(p.dw + (size_t)i * p.K + col, acc[t][a][r]);
            }
        }
    }
}

// dw[co][tap][ci] = sum over the splits of a tile of slab[tile * splits + split][tap][co / 4][ci][co % 4], in split order.
// One thread = one float4 (4 co x 1 ci); a block = 8 co-quads x 32 ci of one (tile, tap): reads of 512 contiguous bytes per 32 lanes
// and split, writes of 128 contiguous bytes per 32 lanes.
__global__ void __launch_bounds__(256) wgrad_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cin, int K,
                                                                int tilesJ, int splits) {
    int b = blockIdx.x;
    const int q8 = b & 7; b >>= 3;                       // group of 8 co-quads
    const int tap = b % 9;
    const int tile = b / 9;
    const int tj = tile % tilesJ, ti = tile / tilesJ;
    const int ci = threadIdx.x & 31, co4 = q8 * 8 + (threadIdx.x >> 5);
    const float* src = slab + (size_t)tile * splits * SLAB_FLOATS + (size_t)((tap * 64 + co4) * TJ + ci) * 4;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < splits; ++s) {
        const floatx4 u = *reinterpret_cast<const floatx4*>(src + (size_t)s * SLAB_FLOATS);
        v += u;
    }
    float* dst = dw + (size_t)(ti * TI + co4 * 4) * K + (size_t)tap * Cin + tj * TJ + ci;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[(size_t)j * K] = v[j];
}

}  // namespace

int64_t conv_wgrad_bf16_slab_bytes() { return (int64_t)MAX_BLOCKS * SLAB_FLOATS * (int64_t)sizeof(float); }

// returns 1 when the launch was taken, 0 when the shape is not covered (caller: conv_wgrad_bf16.hip), < 0 on error
// slab: conv_wgrad_bf16_slab_bytes() of scratch for the partial tiles (dw is then overwritten), or NULL (atomics into the zeroed dw)
int try_conv_wgrad_bf16_img(const uint16_t* x, const uint16_t* dy, float* dw, const ConvGeom& g, unsigned x_bytes, unsigned dy_bytes,
                            hipStream_t stream, float* slab) {
    if (g.R != 3 || g.S != 3 || g.stride != 1 || g.pad != 1 || g.Hin != HW || g.Win != HW || g.Hout != HW || g.Wout != HW) return 0;
    if (g.Cin % TJ != 0 || g.Cout % TI != 0) return 0;
    if (g.Cin > 896) return 0;                           // (the patch-offset table holds image-relative byte offsets / 16 in 16 bits, 0xFFFF = halo)
    if (SIMQ_TUNE_INT("SIMQ_BF16_WGRAD_IMG", 1) == 0) return 0;
    WgradImgArgs p;
    p.x = x; p.dy = dy; p.dw = dw; p.Cin = g.Cin; p.Cout = g.Cout; p.K = g.K(); p.B = g.B;
    p.x_bytes = x_bytes; p.dy_bytes = dy_bytes; p.slab = slab;
    p.tilesI = g.Cout / TI;
    p.tilesJ = g.Cin / TJ;
    const int tiles = p.tilesI * p.tilesJ;
    // one block per CU (157 KB of LDS): tiles x splits = 256 blocks, splits a multiple of 8 so that the blocks of one image range
    // share an XCD; at least two images per block, or the prologue (a patch + three dY steps before the first MFMA) is not amortised
    int splits = 256 / tiles;
    splits = (splits / 8) * 8;
    if (splits < 8) return 0;
    while (splits > 8 && (g.B + splits - 1) / splits < 2) splits -= 8;
    if ((g.B + splits - 1) / splits < SIMQ_TUNE_INT("SIMQ_BF16_WGRAD_IMG_MIN", 2)) return 0;
    p.imgs_per_split = (g.B + splits - 1) / splits;
    splits = (g.B + p.imgs_per_split - 1) / p.imgs_per_split;        // only splits that own an image: every launched block writes its slab
    p.splits = splits;
    note_launch("wgrad_bf16_img");
    prof_launch_begin(1, 2.0 * g.M() * p.Cout * p.K, 2.0 * ((double)g.M() * (g.Cin + g.Cout)) + 4.0 * (double)p.Cout * p.K, stream);
    const dim3 grid((unsigned)(tiles * splits)), block(NW * 64);
    p.dbg = 0;
#ifdef SIMQ_ABLATIONS      // timing ablations (tools/wgrad_check.py): compiled into libsimq_ablate.so only
    p.dbg = SIMQ_TUNE_INT("SIMQ_BF16_WGRAD_IMG_DBG", 0) & (32 | 64);
    switch (SIMQ_TUNE_INT("SIMQ_BF16_WGRAD_IMG_DBG", 0) & 31) {
        case 1: hipLaunchKernelGGL(wgrad_bf16_img_kernel<1>, grid, block, 0, stream, p); break;
        case 8: hipLaunchKernelGGL(wgrad_bf16_img_kernel<8>, grid, block, 0, stream, p); break;
        case 16: hipLaunchKernelGGL(wgrad_bf16_img_kernel<16>, grid, block, 0, stream, p); break;
        case 17: hipLaunchKernelGGL(wgrad_bf16_img_kernel<17>, grid, block, 0, stream, p); break;
        case 9: hipLaunchKernelGGL(wgrad_bf16_img_kernel<9>, grid, block, 0, stream, p); break;
        case 24: hipLaunchKernelGGL(wgrad_bf16_img_kernel<24>, grid, block, 0, stream, p); break;
        default: hipLaunchKernelGGL(wgrad_bf16_img_kernel<0>, grid, block, 0, stream, p);
    }
#else
    hipLaunchKernelGGL(wgrad_bf16_img_kernel<0>, grid, block, 0, stream, p);
#endif
    if (slab) {
        SIMQ_CHECK_LAUNCH();
        hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3((unsigned)(tiles * 9 * 8)), dim3(256), 0, stream, slab, dw, g.Cin, p.K, p.tilesJ, splits);
    }
    prof_launch_end(stream);
    SIMQ_CHECK_LAUNCH();
    return 1;
}

}  // namespace simq
