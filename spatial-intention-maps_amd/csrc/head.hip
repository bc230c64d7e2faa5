// Last layer of the Q-network: nn.Conv2d(32, Cout, 1) with bias (reference networks.py:14,26)
// on the 96x96 upsampled feature map.  Cout is 1 or 2 (envs.py:810,1090), so this is a
// bandwidth-bound per-pixel dot product, not a GEMM: 8 lanes share a pixel (one float4 of
// the 32 input channels each), x is read once with 16-B loads and the Q-map is written
// NCHW because the reference's flat action index is CHW-ordered (envs.py:858).
#include "common.h"

namespace simq {

namespace {

constexpr int MAX_COUT = 4;

template <int CIN>
__global__ void __launch_bounds__(256) head_conv3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ q,
                                                             int B, int HW, int Cout) {
    constexpr int L = CIN / 4;   // lanes per pixel
    const int sub = threadIdx.x % L;
    float4 wv[MAX_COUT];
    for (int co = 0; co < MAX_COUT; ++co)
        wv[co] = co < Cout ? *reinterpret_cast<const float4*>(w + co * CIN + sub * 4) : make_float4(0, 0, 0, 0);
    const size_t total = (size_t)B * HW;
    const size_t ppb = blockDim.x / L;
    for (size_t pix = blockIdx.x * ppb + threadIdx.x / L; pix < total; pix += (size_t)gridDim.x * ppb) {
        float4 v = *reinterpret_cast<const float4*>(x + pix * CIN + sub * 4);
        const unsigned b = (unsigned)pix / (unsigned)HW, p = (unsigned)pix - b * (unsigned)HW;   // 32-bit: B*HW < 2^31
        for (int co = 0; co < Cout; ++co) {
            float s = v.x * wv[co].x + v.y * wv[co].y + v.z * wv[co].z + v.w * wv[co].w;
#pragma unroll
            for (int o = L / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            if (sub == 0) q[((size_t)b * Cout + co) * HW + p] = s + bias[co];
        }
    }
}

template <int CIN>
__global__ void __launch_bounds__(256) head_conv3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dq, float* __restrict__ dx,
                                                             float* dw, float* dbias, int B, int HW, int Cout) {
    constexpr int L = CIN / 4;
    __shared__ float red[256 * 4];
    const int sub = threadIdx.x % L;
    float4 wv[MAX_COUT], gw[MAX_COUT];
    float gb[MAX_COUT];
    for (int co = 0; co < MAX_COUT; ++co) {
        wv[co] = co < Cout ? *reinterpret_cast<const float4*>(w + co * CIN + sub * 4) : make_float4(0, 0, 0, 0);
        gw[co] = make_float4(0, 0, 0, 0);
        gb[co] = 0.f;
    }
    const size_t total = (size_t)B * HW;
    const size_t ppb = blockDim.x / L;
    for (size_t pix = blockIdx.x * ppb + threadIdx.x / L; pix < total; pix += (size_t)gridDim.x * ppb) {
        const unsigned b = (unsigned)pix / (unsigned)HW, p = (unsigned)pix - b * (unsigned)HW;   // 32-bit: B*HW < 2^31
        float4 v = *reinterpret_cast<const float4*>(x + pix * CIN + sub * 4);
        float4 o = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int co = 0; co < MAX_COUT; ++co) {
            if (co < Cout) {
                float g = dq[((size_t)b * Cout + co) * HW + p];
                o.x += g * wv[co].x; o.y += g * wv[co].y; o.z += g * wv[co].z; o.w += g * wv[co].w;
                gw[co].x += g * v.x; gw[co].y += g * v.y; gw[co].z += g * v.z; gw[co].w += g * v.w;
                gb[co] += g;
            }
        }
        *reinterpret_cast<float4*>(dx + pix * CIN + sub * 4) = o;
    }
    // block reduction over the pixel lanes that share `sub`, one output channel at a time
    for (int co = 0; co < Cout; ++co) {
        __syncthreads();
        red[threadIdx.x * 4 + 0] = gw[co].x; red[threadIdx.x * 4 + 1] = gw[co].y;
        red[threadIdx.x * 4 + 2] = gw[co].z; red[threadIdx.x * 4 + 3] = gw[co].w;
        __syncthreads();
        if (threadIdx.x < CIN) {
            int s = threadIdx.x / 4, e = threadIdx.x % 4;
            float a = 0.f;
            for (int t = s; t < 256; t += L) a += red[t * 4 + e];
            unsafeAtomicAdd(dw + co * CIN + threadIdx.x, a);
        }
        __syncthreads();
        red[threadIdx.x] = (sub == 0) ? gb[co] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f;
            for (int t = 0; t < 256; ++t) a += red[t];
            unsafeAtomicAdd(dbias + co, a);
        }
    }
}

}  // namespace

int launch_head_conv3_fwd(const float* x, const float* w, const float* bias, float* q, int B, int HW, int Cin, int Cout,
                          hipStream_t stream) {
    SIMQ_REQUIRE(Cin == 32 && Cout >= 1 && Cout <= MAX_COUT, "head_conv3: Cin=%d Cout=%d unsupported", Cin, Cout);
    SIMQ_REQUIRE((size_t)B * HW < 2147483648ull, "head_conv3: too many pixels for 32-bit indexing");
    size_t blocks = ((size_t)B * HW + 31) / 32;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(head_conv3_fwd_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, bias, q, B, HW, Cout);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_head_conv3_bwd(const float* x, const float* w, const float* dq, float* dx, float* dw, float* dbias, int B,
                          int HW, int Cin, int Cout, hipStream_t stream) {
    SIMQ_REQUIRE(Cin == 32 && Cout >= 1 && Cout <= MAX_COUT, "head_conv3: Cin=%d Cout=%d unsupported", Cin, Cout);
    SIMQ_REQUIRE((size_t)B * HW < 2147483648ull, "head_conv3: too many pixels for 32-bit indexing");
    size_t blocks = ((size_t)B * HW + 31) / 32;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(head_conv3_bwd_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, dq, dx, dw, dbias, B,
                       HW, Cout);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
