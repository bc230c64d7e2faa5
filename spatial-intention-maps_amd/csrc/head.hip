// Last layer of the Q-network: nn.Conv2d(32, Cout, 1) with bias (reference networks.py:14,26)
// on the 96x96 upsampled feature map.  Cout is 1 or 2 (envs.py:810,1090), so this is a
// bandwidth-bound per-pixel dot product, not a GEMM: 8 lanes share a pixel (one float4 of
// the 32 input channels each), x is read once with 16-B loads and the Q-map is written
// NCHW because the reference's flat action index is CHW-ordered (envs.py:858).
//
// The forward pass applies it BEFORE the second bilinear x2 (networks.py:25-26 in the other order): both maps are linear and
// the bilinear weights of a pixel sum to 1, so conv3(upsample(a)) + bias == upsample(conv3(a)) + bias up to fp32 rounding
// -- and the 96x96x32 upsampled activation (151 MB at B = 128: one write, one read, every forward) never exists.  The
// one-hot backward re-interpolates the one pixel per transition it needs; only the dense-dQ backward materialises it.
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
            if (sub == 0) q[((size_t)b * Cout + co) * HW + p] = s + (bias ? bias[co] : 0.f);
        }
    }
}

template <int CIN>
__global__ void __launch_bounds__(256) head_conv3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dq, float* __restrict__ dx,
                                                             float* dw, float* dbias, int B, int HW, int Cout, float* slab_w, float* slab_b) {
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
            if (slab_w) slab_w[(size_t)blockIdx.x * Cout * CIN + co * CIN + threadIdx.x] = a;     // deterministic plans: summed in block order afterwards
            else unsafeAtomicAdd(dw + co * CIN + threadIdx.x, a);
        }
        __syncthreads();
        red[threadIdx.x] = (sub == 0) ? gb[co] : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f;
            for (int t = 0; t < 256; ++t) a += red[t];
            if (slab_b) slab_b[(size_t)blockIdx.x * Cout + co] = a;
            else unsafeAtomicAdd(dbias + co, a);
        }
    }
}

// One-hot upstream gradient: the TD loss touches ONE Q-value per transition (train.py:115,129), so dLoss/dQ has exactly B
// non-zeros.  This launch replaces the dense head_conv3_bwd + upsample2x_bwd pair (and the zero-filled dQ map) by B tiny
// blocks: conv3 backward at the one pixel, then the bilinear x2 transpose scatters its 32 channels onto <= 4 pixels of the
// 48x48 head activation gradient (ds1, zero on entry).  g = clamp(q_sa - y, -1, 1) * grad_scale is the Huber derivative.
__device__ __forceinline__ void lerp2x(int o, int in_size, float scale, int& i0, int& i1, float& l0, float& l1) {
    const float real = scale * (float)o;       // ATen upsample_bilinear2d, align_corners=True (as elementwise.hip lerp_coord)
    i0 = (int)real;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

// Eval-mode head tail in one pass (folded BatchNorm plans: the target net, policy.step): z2 is conv2's output at 24x24 with the folded
// BatchNorm-2 affine map already applied; networks.py:21-26 then is  upsample x2 -> ReLU -> conv3 (-> upsample x2 + bias, head_upsample_q).
// The 48x48x32 activation between the upsample and conv3 (37.7 MB at B = 128: one write, one read) is never stored: 8 lanes share an
// output pixel (one float4 of the 32 channels each), interpolate, rectify and reduce.  Same expressions as upsample2x_fwd_kernel /
// head_conv3_fwd_kernel, so the result is bit-identical to the two launches it replaces.
__global__ void __launch_bounds__(256) head_up_relu_conv3_kernel(const float* __restrict__ z2, const float* __restrict__ w, float* __restrict__ z,
                                                                  int B, int Cout) {
    constexpr int CIN = 32, L = CIN / 4, W1 = 24, W2 = 48;
    const int sub = threadIdx.x % L;
    float4 wv[MAX_COUT];
    for (int co = 0; co < MAX_COUT; ++co)
        wv[co] = co < Cout ? *reinterpret_cast<const float4*>(w + co * CIN + sub * 4) : make_float4(0, 0, 0, 0);
    const float sc = (float)(W1 - 1) / (float)(W2 - 1);
    const unsigned total = (unsigned)B * W2 * W2, ppb = 256 / L;
    for (unsigned pix = blockIdx.x * ppb + threadIdx.x / L; pix < total; pix += gridDim.x * ppb) {
        const unsigned b = pix / (W2 * W2), p = pix - b * (W2 * W2);
        const int oy = (int)(p / W2), ox = (int)(p - (unsigned)oy * W2);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        lerp2x(oy, W1, sc, y0, y1, ly0, ly1);
        lerp2x(ox, W1, sc, x0, x1, lx0, lx1);
        const float* base = z2 + (size_t)b * W1 * W1 * CIN + sub * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(base + (y0 * W1 + x0) * CIN), v01 = *reinterpret_cast<const float4*>(base + (y0 * W1 + x1) * CIN);
        const float4 v10 = *reinterpret_cast<const float4*>(base + (y1 * W1 + x0) * CIN), v11 = *reinterpret_cast<const float4*>(base + (y1 * W1 + x1) * CIN);
        float4 v;
        v.x = fmaxf(ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x), 0.f);
        v.y = fmaxf(ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y), 0.f);
        v.z = fmaxf(ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z), 0.f);
        v.w = fmaxf(ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w), 0.f);
        for (int co = 0; co < Cout; ++co) {
            float s = v.x * wv[co].x + v.y * wv[co].y + v.z * wv[co].z + v.w * wv[co].w;
#pragma unroll
            for (int o = L / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            if (sub == 0) z[((size_t)b * Cout + co) * (W2 * W2) + p] = s;
        }
    }
}

__global__ void __launch_bounds__(32) head_onehot_bwd_kernel(const float* __restrict__ ah2, const float* __restrict__ w3,
                                                             const int64_t* __restrict__ action, const float* __restrict__ q_sa,
                                                             const float* __restrict__ y, float grad_scale, float* ds1, float* dw3,
                                                             float* db3, int Cout, const float* __restrict__ ypre, int ypre_bf16,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd, double* red, int B) {
    constexpr int W2 = 96, W1 = 48, CIN = 32;
    // one block per transition (the default), or ONE block that walks the transitions in order (deterministic plans: the B atomic adds
    // into dw3 / db3 / red then happen in a fixed order)
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int ci = threadIdx.x;
    const int64_t a = action[b];
    const int co = (int)(a / (W2 * W2)), p = (int)(a - (int64_t)co * W2 * W2);
    if (co >= Cout) continue;
    const int oy = p / W2, ox = p - oy * W2;
    const float d = q_sa[b] - y[b];
    const float g = fminf(fmaxf(d, -1.f), 1.f) * grad_scale;
    const float s = (float)(W1 - 1) / (float)(W2 - 1);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lerp2x(oy, W1, s, y0, y1, ly0, ly1);
    lerp2x(ox, W1, s, x0, x1, lx0, lx1);
    // the upsampled activation at this pixel (same expression as upsample2x_fwd_kernel)
    const float* src = ah2 + (size_t)b * W1 * W1 * CIN + ci;
    const float up = ly0 * (lx0 * src[(y0 * W1 + x0) * CIN] + lx1 * src[(y0 * W1 + x1) * CIN]) +
                     ly1 * (lx0 * src[(y1 * W1 + x0) * CIN] + lx1 * src[(y1 * W1 + x1) * CIN]);
    unsafeAtomicAdd(dw3 + co * CIN + ci, g * up);
    if (ci == 0) unsafeAtomicAdd(db3 + co, g);
    const float dx = g * w3[co * CIN + ci];
    float* base = ds1 + (size_t)b * W1 * W1 * CIN + ci;      // this thread owns channel ci of sample b: plain read-modify-write
    base[(y0 * W1 + x0) * CIN] += ly0 * lx0 * dx;
    base[(y0 * W1 + x1) * CIN] += ly0 * lx1 * dx;
    base[(y1 * W1 + x0) * CIN] += ly1 * lx0 * dx;
    base[(y1 * W1 + x1) * CIN] += ly1 * lx1 * dx;
    // The gradient of this sample is zero outside these <= 4 pixels, so the BatchNorm-backward sums of the layer in front (sum dz,
    // sum dz * xhat with dz = ds1 * [a > 0]) are complete after a look at them -- no pass over the 48x48x32 map (red == NULL: not fused)
    if (!red) continue;
    const float mu = mean[ci], is = invstd[ci];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (a == 1 && y1 == y0) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 1 && x1 == x0) continue;
            const size_t o = ((size_t)b * W1 * W1 + (a ? y1 : y0) * W1 + (e ? x1 : x0)) * CIN + ci;
            const float dz = ah2[o] > 0.f ? ds1[o] : 0.f;
            const float yv = ypre_bf16 ? __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(ypre)[o] << 16) : ypre[o];
            s0 += dz;
            s1 += dz * ((yv - mu) * is);
        }
    }
    unsafeAtomicAdd(red + ci, (double)s0);
    unsafeAtomicAdd(red + CIN + ci, (double)s1);
    }
}

// q[b][co] (96x96, NCHW) = bilinear x2 (align_corners=True) of z[b][co] (48x48) + bias[co]: 4 outputs along x per thread
__global__ void __launch_bounds__(256) head_upsample_q_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              float* __restrict__ q, int planes, int Cout, int vec) {
    constexpr int W2 = 96, W1 = 48;
    const float s = (float)(W1 - 1) / (float)(W2 - 1);
    const unsigned total = (unsigned)planes * W2 * (W2 / 4);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned x4 = i % (W2 / 4), r = i / (W2 / 4);
        const unsigned oy = r % W2, pl = r / W2;
        const float bv = bias[pl % (unsigned)Cout];
        int y0, y1; float ly0, ly1;
        lerp2x((int)oy, W1, s, y0, y1, ly0, ly1);
        const float* r0 = z + (size_t)pl * W1 * W1 + y0 * W1;
        const float* r1 = z + (size_t)pl * W1 * W1 + y1 * W1;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int x0, x1; float lx0, lx1;
            lerp2x((int)x4 * 4 + e, W1, s, x0, x1, lx0, lx1);
            o[e] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]) + bv;
        }
        float* dst = q + ((size_t)pl * W2 + oy) * W2 + x4 * 4;
        if (vec) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else { dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3]; }
    }
}

}  // namespace

int launch_head_upsample_q(const float* z, const float* bias, float* q, int B, int Cout, hipStream_t stream) {
    SIMQ_REQUIRE(Cout >= 1 && Cout <= MAX_COUT, "head_upsample_q: Cout=%d unsupported", Cout);
    SIMQ_REQUIRE((size_t)B * Cout * 9216 < 2147483648ull, "head_upsample_q: too many pixels for 32-bit indexing");
    const unsigned total = (unsigned)B * Cout * 96 * 24;
    unsigned blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(head_upsample_q_kernel, dim3(blocks), dim3(256), 0, stream, z, bias, q, B * Cout, Cout,
                       (reinterpret_cast<uintptr_t>(q) & 15) == 0 ? 1 : 0);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_head_onehot_bwd(const float* ah2, const float* w3, const int64_t* action, const float* q_sa, const float* y,
                           float grad_scale, float* ds1, float* dw3, float* db3, int B, int Cout, hipStream_t stream,
                           const float* ypre, int ypre_bf16, const float* mean, const float* invstd, double* red, int serial) {
    SIMQ_REQUIRE(Cout >= 1 && Cout <= MAX_COUT, "head_onehot_bwd: Cout=%d unsupported", Cout);
    SIMQ_CHECK_HIP(hipMemsetAsync(ds1, 0, sizeof(float) * (size_t)B * 48 * 48 * 32, stream));
    hipLaunchKernelGGL(head_onehot_bwd_kernel, dim3(serial ? 1 : B), dim3(32), 0, stream, ah2, w3, action, q_sa, y, grad_scale, ds1, dw3, db3, Cout,
                       ypre, ypre_bf16, mean, invstd, red, B);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_head_up_relu_conv3(const float* z2, const float* w, float* z, int B, int Cout, hipStream_t stream) {
    SIMQ_REQUIRE(Cout >= 1 && Cout <= MAX_COUT, "head_up_relu_conv3: Cout=%d unsupported", Cout);
    SIMQ_REQUIRE((size_t)B * 2304 < 2147483648ull / 32, "head_up_relu_conv3: too many pixels for 32-bit indexing");
    size_t blocks = ((size_t)B * 2304 + 31) / 32;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(head_up_relu_conv3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z2, w, z, B, Cout);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

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
                          int HW, int Cin, int Cout, hipStream_t stream, float* det_slab) {
    SIMQ_REQUIRE(Cin == 32 && Cout >= 1 && Cout <= MAX_COUT, "head_conv3: Cin=%d Cout=%d unsupported", Cin, Cout);
    SIMQ_REQUIRE((size_t)B * HW < 2147483648ull, "head_conv3: too many pixels for 32-bit indexing");
    size_t blocks = ((size_t)B * HW + 31) / 32;
    if (blocks > 1024) blocks = 1024;
    float* slab_w = det_slab;
    float* slab_b = det_slab ? det_slab + 1024 * MAX_COUT * 32 : nullptr;
    hipLaunchKernelGGL(head_conv3_bwd_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, stream, x, w, dq, dx, dw, dbias, B,
                       HW, Cout, slab_w, slab_b);
    SIMQ_CHECK_LAUNCH();
    if (det_slab) {                                   // (dw / dbias are overwritten: they were zero, nothing else adds into them)
        if (int rc = launch_wgrad_slab_sum(slab_w, dw, (int64_t)Cout * 32, (int)blocks, stream)) return rc;
        return launch_wgrad_slab_sum(slab_b, dbias, Cout, (int)blocks, stream);
    }
    return 0;
}

}  // namespace simq
