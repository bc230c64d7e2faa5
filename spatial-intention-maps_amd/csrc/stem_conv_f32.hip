// First convolution of the encoder (reference resnet.py:94: conv 7x7, stride 2, pad 3, Cin = 3..9 -> 64, no bias) in exact fp32 on the
// matrix cores (v_mfma_f32_16x16x4_f32), for fp32 / split-bf16 plans.  With 3..9 input channels the implicit-GEMM kernel cannot use its
// vector loader (Cin % 16) and gathers the im2col matrix one float at a time: 96 us per launch at B = 32, three launches per step.
//
// The structure of stem_conv_bf16.hip in fp32:
//   * In NHWC the 7 taps x Cin channels of one filter ROW are 7*Cin CONTIGUOUS floats of x, starting at pixel (2 oy + ky - 3, 2 ox - 3) --
//     and in OHWI they are 7*Cin contiguous floats of w as well.  Per filter row the contraction runs over K = 7*Cin, in "super-steps"
//     of 16 floats: lane (pixel i, k-group g) loads the FOUR consecutive floats 16 s + 4 g .. + 3 of its pixel's run straight from
//     global memory (one 16-byte load, dword-aligned), the weight fragment of lane (channel n, g) is the same four floats of row n in
//     LDS (one ds_read_b128), and component c of both feeds MFMA c of the super-step: the K index of a 16x16x4 MFMA is free as long
//     as A and B agree, so logical k = 16 s + 4 g + c needs no shuffling.  16 MFMAs per 16-pixel tile and super-step (4 channel tiles).
//   * Left / right padding: elements outside the image row are masked (only when a wave-level ballot says some lane touches them);
//     rows above / below the image are skipped; runs that would leave the tensor at its two ends are loaded element-wise.
//   * The weights (64 x 7 rows, each padded to 16 * ceil(7 Cin / 16) floats with zeros; row pitch + 4 floats: conflict-free
//     ds_read_b128) are copied from the OHWI parameters into LDS once per persistent block -- no derived weight copy is kept;
//     the BatchNorm batch statistics of a block leave with one fp64 atomic per channel and block.
//   * One wave = one 16-pixel tile at a time, nine waves per block, two blocks per CU: at B = 32 the 4608 tiles are exactly one per
//     wave of 512 resident blocks (4.5 waves per SIMD), and the global-load latency of a wave hides behind the MFMAs of its neighbours.
// Arithmetic: exact fp32 FMA chains (the fp32 parity bars apply unchanged); only the summation order differs from the generic kernel.
#include <cstdint>

#include <atomic>

#include "common.h"

namespace simq {

namespace {

typedef float floatx4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte load the compiler may not assume aligned

constexpr int COUT = 64, R = 7, NWAVE = 9;     // 9 waves x 512 blocks = the 4608 tiles of B = 32, one each, 4.5 waves per SIMD

struct StemF32Args {
    const float* x; const float* w; float* y; double* stats;
    int B, H, W, C, Ho, Wo;
    int krowf, wrowf;                 // floats per padded filter row, LDS pitch of one output channel
};

// NSS = super-steps per filter row = ceil(7 C / 16)
template <int NSS>
__global__ void __launch_bounds__(NWAVE * 64, 2) stem_conv_f32_kernel(const StemF32Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wl = reinterpret_cast<float*>(smem);
    double* red = reinterpret_cast<double*>(smem + (size_t)COUT * p.wrowf * sizeof(float));          // [NWAVE][64][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kreal = R * p.C;                                       // real k per filter row
    constexpr int KROWF = 16 * NSS;
    // weights -> LDS: [o][ky][KROWF] with zeros behind the 7*C real values (OHWI: the 7*C floats of (o, ky) are contiguous).  All the
    // loads of a thread are issued before its first store (one element at a time the fill is a chain of global load latencies and
    // dominated the launch); 16-byte pieces when 7*C is a multiple of 4.
    if ((p.C & 3) == 0) {
        constexpr int V = KROWF / 4, NV = COUT * R * V, TRIPS = (NV + NWAVE * 64 - 1) / (NWAVE * 64);
        floatx4 v[TRIPS];
#pragma unroll
        for (int u = 0; u < TRIPS; ++u) {
            const int i = tid + u * NWAVE * 64;
            const int row = i / V, n = (i - row * V) * 4;            // row = o * 7 + ky
            v[u] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (i < NV && n < kreal) v[u] = *reinterpret_cast<const floatx4*>(p.w + (size_t)row * kreal + n);
        }
#pragma unroll
        for (int u = 0; u < TRIPS; ++u) {
            const int i = tid + u * NWAVE * 64;
            const int row = i / V, n = (i - row * V) * 4;
            const int o = row / R, ky = row - o * R;
            if (i < NV) *reinterpret_cast<floatx4*>(wl + o * p.wrowf + ky * KROWF + n) = v[u];
        }
    } else {
        constexpr int NE = COUT * R * KROWF;
        for (int i0 = tid; i0 < NE; i0 += 8 * NWAVE * 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NWAVE * 64;
                const int row = i / KROWF, n = i - row * KROWF;
                v[u] = (i < NE && n < kreal) ? p.w[(size_t)row * kreal + n] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * NWAVE * 64;
                const int row = i / KROWF, n = i - row * KROWF;
                const int o = row / R, ky = row - o * R;
                if (i < NE) wl[o * p.wrowf + ky * KROWF + n] = v[u];
            }
        }
    }
    __syncthreads();

    const long total = (long)p.B * p.H * p.W * p.C;                  // floats in x
    const int fi = lane & 15, kg = lane >> 4;
    const int rowf = p.W * p.C;                                      // floats per image row
    const int tiles_per_row = p.Wo / 16;
    const int ntiles = p.B * p.Ho * tiles_per_row;                   // 16-pixel tiles
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    const float* wfrag = wl + fi * p.wrowf + kg * 4;                 // + j * 16 * wrowf + ky * KROWF + s * 16

    // One wave = one 16-pixel x 64-channel tile at a time; the load latency of a wave hides behind the MFMAs of the other waves of its SIMD
    for (int tile = blockIdx.x * NWAVE + wave; tile < ntiles; tile += gridDim.x * NWAVE) {
        floatx4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const int ox0 = (tile % tiles_per_row) * 16;
        const int r = tile / tiles_per_row;
        const int oy = r % p.Ho, b = r / p.Ho;
        const int e_lo = (2 * (ox0 + fi) - 3) * p.C + kg * 4;         // element of the image row this lane's run starts at (super-step 0)
        const float* prow = p.x + ((long)(b * p.H + 2 * oy - 3) * rowf + e_lo);
        // some lane of the wave touches the left / right padding in super-step s (wave-uniform)
        bool side[NSS];
#pragma unroll
        for (int s = 0; s < NSS; ++s) { const int e = e_lo + s * 16; side[s] = __builtin_amdgcn_ballot_w64(e < 0 || e + 4 > rowf) != 0; }
#pragma unroll 1
        for (int ky = 0; ky < R; ++ky) {
            const int iy = 2 * oy + ky - 3;
            if ((unsigned)iy >= (unsigned)p.H) continue;                                      // wave-uniform: a padding row
            const bool tensor_edge = (b == 0 && iy == 0) || (b == p.B - 1 && iy == p.H - 1);  // wave-uniform
            floatx4 av[NSS];
            // fragment (filter row ky, super-step s): four consecutive floats of the lane's run
#pragma unroll
            for (int s = 0; s < NSS; ++s) {
                av[s] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (s * 16 + kg * 4 < kreal) {
                    const float* src = prow + (ky * rowf + s * 16);
                    if (!tensor_edge) {
                        const floatx4_a4 u = *reinterpret_cast<const floatx4_a4*>(src);
                        av[s] = floatx4{u[0], u[1], u[2], u[3]};
                    } else {
                        const long g0 = src - p.x;
#pragma unroll
                        for (int q = 0; q < 4; ++q) av[s][q] = (g0 + q >= 0 && g0 + q < total) ? src[q] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < NSS; ++s) {
                if (side[s]) {
                    const int e = e_lo + s * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) av[s][q] = ((unsigned)(e + q) < (unsigned)rowf) ? av[s][q] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(wfrag + j * 16 * p.wrowf + ky * KROWF + s * 16);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][c], bv[c], acc[j], 0, 0, 0);
                }
            }
        }
        // ---- store + statistics: lane holds column fi of N-tile j, rows 4 kg + r of the tile
        const size_t m0 = (size_t)tile * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float v = acc[j][r4];
                s0[j] += v; s1[j] += v * v;
                p.y[(m0 + 4 * kg + r4) * COUT + j * 16 + fi] = v;
            }
    }
    if (!p.stats) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = s0[j], c = s1[j];
        a += __shfl_xor(a, 16); c += __shfl_xor(c, 16);
        a += __shfl_xor(a, 32); c += __shfl_xor(c, 32);
        if (kg == 0) { red[(wave * COUT + j * 16 + fi) * 2] = (double)a; red[(wave * COUT + j * 16 + fi) * 2 + 1] = (double)c; }
    }
    __syncthreads();
    if (tid < COUT) {
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) { a += red[(w * COUT + tid) * 2]; c += red[(w * COUT + tid) * 2 + 1]; }
        unsafeAtomicAdd(p.stats + tid, a);
        unsafeAtomicAdd(p.stats + COUT + tid, c);
    }
}

}  // namespace

bool stem_conv_f32_eligible(int H, int W, int C, int cout, int k, int stride, int pad) {
    return cout == COUT && k == R && stride == 2 && pad == 3 && C >= 1 && R * C <= 64 && H % 2 == 0 && W % 32 == 0;
}

// y [B][H/2][W/2][64] = conv7x7 s2 p3 of x [B][H][W][C] with w [64][7][7][C] (OHWI); stats (may be NULL): [sum | sum of squares] += per channel
int launch_stem_conv_f32(const float* x, const float* w_ohwi, float* y, double* stats, int B, int H, int W, int C, hipStream_t stream) {
    SIMQ_REQUIRE(stem_conv_f32_eligible(H, W, C, COUT, R, 2, 3), "stem_conv_f32: shape %dx%dx%d not covered", H, W, C);
    SIMQ_REQUIRE((double)B * H * W * C < 2147483000.0, "stem_conv_f32: input too large for 32-bit element indexing");
    StemF32Args p;
    p.x = x; p.w = w_ohwi; p.y = y; p.stats = stats;
    p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = H / 2; p.Wo = W / 2;
    const int nss = (R * C + 15) / 16;
    p.krowf = 16 * nss;
    p.wrowf = R * p.krowf + 4;                                        // 228 floats at C <= 4: 36 mod 64 banks
    const int ntiles = B * p.Ho * (p.Wo / 16);
    const int smem = COUT * p.wrowf * (int)sizeof(float) + NWAVE * COUT * 2 * (int)sizeof(double);
    const int per_cu = smem <= 78 * 1024 ? 2 : 1;
    int blocks = (ntiles + NWAVE - 1) / NWAVE;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;                 // persistent blocks
    void (*kern)(const StemF32Args) = nss == 1 ? stem_conv_f32_kernel<1> : nss == 2 ? stem_conv_f32_kernel<2>
                                    : nss == 3 ? stem_conv_f32_kernel<3> : stem_conv_f32_kernel<4>;
    // (the attribute is per device and per function: remembered per device, and a racing second thread only repeats the call)
    static std::atomic<int> attr_smem[64][5];
    int dev = 0;
    SIMQ_CHECK_HIP(hipGetDevice(&dev));
    SIMQ_REQUIRE(dev >= 0 && dev < 64, "stem_conv_f32: device index %d out of range", dev);
    if (smem > attr_smem[dev][nss].load(std::memory_order_relaxed)) {
        SIMQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_smem[dev][nss].store(smem, std::memory_order_relaxed);
    }
    note_launch("stem_conv_f32");
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NWAVE * 64), smem, stream, p);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
