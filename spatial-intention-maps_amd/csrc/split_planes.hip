// bf16 plane producers for the matrix-core convolution paths (conv_igemm_bf16.hip / conv_wgrad_bf16.hip).
//   hi = bf16(v) (round-to-nearest-even), lo = bf16(v - float(hi))   -> v ~= hi + lo to ~2^-17 relative.
// Standalone splitters (weights once per optimiser step; generic tensors in the unit tests); the activation /
// gradient planes of the network are written by the fused elementwise kernels (elementwise.hip) instead.
#include "common.h"

namespace simq {

namespace {

__device__ __forceinline__ uint16_t to_bf16(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float from_bf16(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

__global__ void split_planes_kernel(const float* __restrict__ src, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = *reinterpret_cast<const float4*>(src + i * 4);
        ushort4 h = make_ushort4(to_bf16(v.x), to_bf16(v.y), to_bf16(v.z), to_bf16(v.w));
        *reinterpret_cast<ushort4*>(hi + i * 4) = h;
        if (lo) {
            ushort4 l = make_ushort4(to_bf16(v.x - from_bf16(h.x)), to_bf16(v.y - from_bf16(h.y)),
                                     to_bf16(v.z - from_bf16(h.z)), to_bf16(v.w - from_bf16(h.w)));
            *reinterpret_cast<ushort4*>(lo + i * 4) = l;
        }
    }
}

// weights: planes of w (OHWI) and of the flipped/transposed copy wt[ci][taps-1-t][co] used by dgrad
__global__ void weight_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ w_hi, uint16_t* __restrict__ w_lo,
                                     uint16_t* __restrict__ wt_hi, uint16_t* __restrict__ wt_lo, int cout, int taps, int cin) {
    size_t total = (size_t)cout * taps * cin;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        // i indexes the transposed layout (co fastest) so that its writes coalesce; the OHWI planes are written from
        // the same thread at the source index (strided, but this kernel moves 45 MB once per step)
        int co = (int)(i % cout);
        size_t r = i / cout;
        int tf = (int)(r % taps);
        int ci = (int)(r / taps);
        size_t src = ((size_t)co * taps + (taps - 1 - tf)) * cin + ci;
        float v = w[src];
        uint16_t h = to_bf16(v);
        uint16_t l = to_bf16(v - from_bf16(h));
        w_hi[src] = h;
        if (w_lo) w_lo[src] = l;
        if (wt_hi) {
            wt_hi[i] = h;
            if (wt_lo) wt_lo[i] = l;
        }
    }
}

// All convolutions of the network in ONE launch (blockIdx.y = convolution): per step this replaces 21-22 tiny launches.
// For every weight element: optional fp32 flipped/transposed copy (fp32 dgrad), optional bf16 planes of the OHWI
// weight and of the flipped/transposed weight (matrix-core precisions).
__global__ void weight_prep_all_kernel(const float* __restrict__ params, WeightPrepTable t, float* __restrict__ wt_f32,
                                       uint16_t* __restrict__ wpl, uint16_t* __restrict__ wtpl, int np, int64_t wp_total) {
    const WeightPrepDesc d = t.d[blockIdx.y];
    const float* w = params + d.w_off;
    const size_t total = (size_t)d.cout * d.taps * d.cin;
    if (blockIdx.z == 0) {          // OHWI order: 16-B reads, 8-B plane writes (weights are 16-B aligned in the flat buffer)
        if (!wpl) return;
        const size_t total4 = total / 4;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
            const float4 v = *reinterpret_cast<const float4*>(w + i * 4);
            const ushort4 h = make_ushort4(to_bf16(v.x), to_bf16(v.y), to_bf16(v.z), to_bf16(v.w));
            *reinterpret_cast<ushort4*>(wpl + d.wp_off + i * 4) = h;
            if (np == 2)
                *reinterpret_cast<ushort4*>(wpl + wp_total + d.wp_off + i * 4) =
                    make_ushort4(to_bf16(v.x - from_bf16(h.x)), to_bf16(v.y - from_bf16(h.y)), to_bf16(v.z - from_bf16(h.z)),
                                 to_bf16(v.w - from_bf16(h.w)));
        }
        return;
    }
    if (!wt_f32 && !wtpl) return;
    // flipped / transposed order wt[ci][taps-1-t][co] = w[co][t][ci]: one 64(co) x 64(ci) tile of one tap per block pass,
    // transposed through LDS so that both the reads (along ci) and the writes (along co) are coalesced
    __shared__ float tile[64][65];
    const int tiles_co = (d.cout + 63) / 64, tiles_ci = (d.cin + 63) / 64;
    const int ntiles = tiles_co * tiles_ci * d.taps;
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
        const int tap = bid % d.taps;
        const int rest = bid / d.taps;
        const int ci0 = (rest % tiles_ci) * 64, co0 = (rest / tiles_ci) * 64;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int co = co0 + y + 4 * k, ci = ci0 + x;
            tile[y + 4 * k][x] = (co < d.cout && ci < d.cin) ? w[((size_t)co * d.taps + tap) * d.cin + ci] : 0.f;
        }
        __syncthreads();
        const int tf = d.taps - 1 - tap;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ci = ci0 + y + 4 * k, co = co0 + x;
            if (ci < d.cin && co < d.cout) {
                const float v = tile[x][y + 4 * k];
                const size_t i = ((size_t)ci * d.taps + tf) * d.cout + co;
                if (wt_f32) wt_f32[d.wt_off + i] = v;
                if (wtpl) {
                    const uint16_t h = to_bf16(v);
                    wtpl[d.wp_off + i] = h;
                    if (np == 2) wtpl[wp_total + d.wp_off + i] = to_bf16(v - from_bf16(h));
                }
            }
        }
    }
}

}  // namespace

int launch_weight_prep_all(const float* params, const WeightPrepTable& t, float* wt_f32, uint16_t* wpl, uint16_t* wtpl, int np,
                           int64_t wp_total, hipStream_t stream) {
    if (t.n == 0) return 0;
    hipLaunchKernelGGL(weight_prep_all_kernel, dim3(192, t.n, 2), dim3(256), 0, stream, params, t, wt_f32, wpl, wtpl, np, wp_total);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_split_planes(const float* src, uint16_t* hi, uint16_t* lo, int64_t n, hipStream_t stream) {
    SIMQ_REQUIRE(n % 4 == 0, "split_planes: n must be a multiple of 4");
    size_t n4 = (size_t)n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, hi, lo, n4);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_weight_planes(const float* w, uint16_t* w_hi, uint16_t* w_lo, uint16_t* wt_hi, uint16_t* wt_lo, int cout, int taps,
                         int cin, hipStream_t stream) {
    size_t total = (size_t)cout * taps * cin;
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(weight_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w, w_hi, w_lo, wt_hi, wt_lo, cout,
                       taps, cin);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
