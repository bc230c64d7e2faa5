// Channel-last (NHWC) elementwise and per-channel reduction kernels for gfx950.
// All are HBM-bound: 16-B vector accesses along C, grid-stride, fp64 only for the
// cross-block combination of per-channel sums.
//
// Reference operators replaced (PyTorch modules used by networks.py / resnet.py):
//   nn.BatchNorm2d train/eval forward + backward   resnet.py:24,27,57,82 ; networks.py:11,13
//   nn.ReLU / residual add                          resnet.py:25,44-45
//   nn.MaxPool2d(3, 2, 1) forward + backward        resnet.py:59
//   F.interpolate(x2, bilinear, align_corners=True) networks.py:21,25 (+ backward)
//   ToTensor HWC->CHW of policies.py:44-45 (layout helpers)
#include "common.h"
#include "bn_coeff.h"

namespace simq {

namespace {

// eval mode: scale / shift of every BatchNorm layer from the running statistics in ONE launch (one block per layer); the
// convolution epilogues then apply them (+ residual + ReLU) and no bn_apply launch is needed.
__global__ void bn_eval_coeff_kernel(BnEvalTable t, const float* __restrict__ params, const float* __restrict__ bnbuf,
                                     float* __restrict__ aux) {
    const BnEvalDesc d = t.d[blockIdx.x];
    BnRef b;
    b.gamma = params + d.g_off; b.beta = params + d.b_off;
    b.rmean = const_cast<float*>(bnbuf) + d.buf_off; b.rvar = const_cast<float*>(bnbuf) + d.buf_off + d.C;
    b.C = d.C;
    for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
        float sc, sh, m, i; double md, vd;
        bn_coeff(b, c, sc, sh, m, i, md, vd);
        aux[d.aux_off + c] = sc;
        aux[d.aux_off + d.C + c] = sh;
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 fma4(float4 a, float4 s, float4 t) {
    return make_float4(fmaf(a.x, s.x, t.x), fmaf(a.y, s.y, t.y), fmaf(a.z, s.z, t.z), fmaf(a.w, s.w, t.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// optional bf16 plane outputs (hi = bf16(v), lo = bf16(v - hi)) for the matrix-core convolution paths
__device__ __forceinline__ uint16_t to_bf16(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ __forceinline__ float from_bf16(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
// read 4 values back from bf16 planes (hi [+ lo]); mask form: > 0 of a bf16 value == positive non-zero as int16
__device__ __forceinline__ float4 ld_planes(const Planes& pl, size_t i4) {
    const ushort4 h = *reinterpret_cast<const ushort4*>(pl.hi + i4 * 4);
    float4 v = make_float4(from_bf16(h.x), from_bf16(h.y), from_bf16(h.z), from_bf16(h.w));
    if (pl.lo) {
        const ushort4 l = *reinterpret_cast<const ushort4*>(pl.lo + i4 * 4);
        v.x += from_bf16(l.x); v.y += from_bf16(l.y); v.z += from_bf16(l.z); v.w += from_bf16(l.w);
    }
    return v;
}
__device__ __forceinline__ void st_planes(const Planes& pl, size_t i4, float4 v) {
    if (!pl.hi) return;
    ushort4 h = make_ushort4(to_bf16(v.x), to_bf16(v.y), to_bf16(v.z), to_bf16(v.w));
    *reinterpret_cast<ushort4*>(pl.hi + i4 * 4) = h;
    if (pl.lo) {
        ushort4 l = make_ushort4(to_bf16(v.x - from_bf16(h.x)), to_bf16(v.y - from_bf16(h.y)),
                                 to_bf16(v.z - from_bf16(h.z)), to_bf16(v.w - from_bf16(h.w)));
        *reinterpret_cast<ushort4*>(pl.lo + i4 * 4) = l;
    }
}
// a pre-BatchNorm convolution output: fp32, or bf16 behind the same pointer (plain-bf16 plans, ConvEpilogue::y_bf16)
__device__ __forceinline__ float4 ld4y(const float* y, size_t i4, int y_bf16) {
    if (!y_bf16) return ld4(y + i4 * 4);
    const ushort4 h = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(y) + i4 * 4);
    return make_float4(from_bf16(h.x), from_bf16(h.y), from_bf16(h.z), from_bf16(h.w));
}
__device__ __forceinline__ float4 relu4(float4 a) {
    return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
}

// out = [relu]( y*scale+shift [+ res | + res*rscale+rshift] ), one float4 per thread-iteration
// res / res_pl: the residual as fp32 or (matrix-core precisions, where the fp32 copy of a block activation is not kept) as its
// bf16 planes; out may be NULL when only the planes are consumed downstream
__global__ void bn_apply_kernel(const float* __restrict__ y, BnRef bn, const float* __restrict__ res, Planes res_pl, BnRef rbn, int has_rbn,
                                int relu, float* __restrict__ out, Planes pl, size_t total4, int C4, int y_bf16) {
    // the grid stride (gridDim*blockDim) is a multiple of C4, so a thread keeps its 4 channels: coefficients once
    const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const int c = (int)(i0 % C4) * 4;
    float4 sc, sh, rsc, rsh;
    if (bn.C <= kCoeffMaxC) {
        __shared__ float cs[4 * kCoeffMaxC];
        bn_coeff_block(bn, cs);
        if (has_rbn) bn_coeff_block(rbn, cs + 2 * kCoeffMaxC);
        __syncthreads();
        sc = ld4(cs + c); sh = ld4(cs + kCoeffMaxC + c);
        if (has_rbn) { rsc = ld4(cs + 2 * kCoeffMaxC + c); rsh = ld4(cs + 3 * kCoeffMaxC + c); }
    } else {
        bn_coeff4(bn, c, sc, sh);
        if (has_rbn) bn_coeff4(rbn, c, rsc, rsh);
    }
    for (size_t i = i0; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = fma4(ld4y(y, i, y_bf16), sc, sh);
        if (res || res_pl.hi) {
            float4 r = res ? ld4(res + i * 4) : ld_planes(res_pl, i);
            if (has_rbn) r = fma4(r, rsc, rsh);
            v = add4(v, r);
        }
        if (relu) v = relu4(v);
        if (out) st4(out + i * 4, v);
        st_planes(pl, i, v);
    }
    bn_commit(bn);
    if (has_rbn) bn_commit(rbn);
}

// ---- all-bf16 forms (plain-bf16 plans, planes-only activations): 8 channels = one 16-byte access per thread and tensor, two
// grid-stride iterations in flight.  With 8-byte accesses these kernels sat at ~3.2 TB/s -- too few bytes in flight per CU.
struct F8 { float v[8]; };
__device__ __forceinline__ F8 unpack8(uint4 r) {
    F8 o;
    o.v[0] = __uint_as_float(r.x << 16); o.v[1] = __uint_as_float(r.x & 0xffff0000u);
    o.v[2] = __uint_as_float(r.y << 16); o.v[3] = __uint_as_float(r.y & 0xffff0000u);
    o.v[4] = __uint_as_float(r.z << 16); o.v[5] = __uint_as_float(r.z & 0xffff0000u);
    o.v[6] = __uint_as_float(r.w << 16); o.v[7] = __uint_as_float(r.w & 0xffff0000u);
    return o;
}
__device__ __forceinline__ uint4 pack8(const F8& a) {
    uint4 r;
    r.x = (uint32_t)to_bf16(a.v[0]) | ((uint32_t)to_bf16(a.v[1]) << 16); r.y = (uint32_t)to_bf16(a.v[2]) | ((uint32_t)to_bf16(a.v[3]) << 16);
    r.z = (uint32_t)to_bf16(a.v[4]) | ((uint32_t)to_bf16(a.v[5]) << 16); r.w = (uint32_t)to_bf16(a.v[6]) | ((uint32_t)to_bf16(a.v[7]) << 16);
    return r;
}
__device__ __forceinline__ uint4 ld16(const uint16_t* p, size_t i8) { return *reinterpret_cast<const uint4*>(p + i8 * 8); }
__device__ __forceinline__ void st16(uint16_t* p, size_t i8, uint4 v) { *reinterpret_cast<uint4*>(p + i8 * 8) = v; }

__global__ void __launch_bounds__(256) bn_apply16_kernel(const uint16_t* __restrict__ y, BnRef bn, const uint16_t* __restrict__ res, BnRef rbn,
                                                         int has_rbn, int relu, uint16_t* __restrict__ out, size_t total8, int C8) {
    const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const int c = (int)(i0 % C8) * 8;                 // (the grid stride is a multiple of C8: a thread keeps its 8 channels)
    __shared__ float cs[4 * kCoeffMaxC];              // (the launcher bounds C)
    bn_coeff_block(bn, cs);
    if (has_rbn) bn_coeff_block(rbn, cs + 2 * kCoeffMaxC);
    __syncthreads();
    float scp[8], shp[8], rscp[8], rshp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        scp[k] = cs[c + k]; shp[k] = cs[kCoeffMaxC + c + k];
        rscp[k] = has_rbn ? cs[2 * kCoeffMaxC + c + k] : 1.f; rshp[k] = has_rbn ? cs[3 * kCoeffMaxC + c + k] : 0.f;
    }
    auto one = [&](uint4 yr, uint4 rr) {
        F8 a = unpack8(yr), r = unpack8(rr);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = fmaf(a.v[k], scp[k], shp[k]);
            if (res) v += has_rbn ? fmaf(r.v[k], rscp[k], rshp[k]) : r.v[k];
            a.v[k] = relu ? fmaxf(v, 0.f) : v;
        }
        return pack8(a);
    };
    const uint4 z = make_uint4(0, 0, 0, 0);
    size_t i = i0;
    for (; i + stride < total8; i += 2 * stride) {
        const uint4 y0 = ld16(y, i), y1 = ld16(y, i + stride);
        const uint4 r0 = res ? ld16(res, i) : z, r1 = res ? ld16(res, i + stride) : z;
        st16(out, i, one(y0, r0));
        st16(out, i + stride, one(y1, r1));
    }
    if (i < total8) st16(out, i, one(ld16(y, i), res ? ld16(res, i) : z));
    bn_commit(bn);
    if (has_rbn) bn_commit(rbn);
}

// MFY: no mask plane -- the ReLU mask is recomputed from the pre-BN output, (y * mscale + mshift > 0) (Ctx::mask1_from_y: one plane less to
// read).  A template parameter, not a run-time test: a load under a branch is waited for inside its branch, which serialised the three
// streams of this HBM-bound kernel (46 us instead of 27 per launch when it was a run-time test).
template <bool MFY>
__global__ void __launch_bounds__(256) bn_bwd_apply16_kernel(const uint16_t* __restrict__ g, const uint16_t* __restrict__ mask16,
                                                             const uint16_t* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const double* __restrict__ red, uint16_t* __restrict__ dy,
                                                             uint16_t* __restrict__ dz_out, float* dgamma, float* dbeta, size_t total8, int C8,
                                                             float inv_rows, float dparam_scale, const float* __restrict__ mscale,
                                                             const float* __restrict__ mshift) {
    const int C = C8 * 8;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dbeta[c] = (float)red[c] * dparam_scale;
            dgamma[c] = (float)red[C + c] * dparam_scale;
        }
    }
    const size_t i0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const int c = (int)(i0 % C8) * 8;
    // the expression of bn_bwd_apply_kernel with the per-channel factors (gamma*invstd, dbeta/rows, dgamma/rows) formed once
    float ka[8], mu[8], is[8], db[8], dg[8], ms[MFY ? 8 : 1], mh[MFY ? 8 : 1];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mu[k] = mean[c + k]; is[k] = invstd[c + k]; ka[k] = gamma[c + k] * is[k];
        db[k] = (float)red[c + k] * inv_rows; dg[k] = (float)red[C + c + k] * inv_rows;
        if constexpr (MFY) { ms[k] = mscale[c + k]; mh[k] = mshift[c + k]; }
    }
    auto one = [&](uint4 gr, uint4 mr, uint4 yr, uint4& dzr) {
        F8 gv = unpack8(gr), yv = unpack8(yr), o;
        const uint32_t mw[4] = {mr.x, mr.y, mr.z, mr.w};
        uint32_t zw[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const short m = (short)(k & 1 ? mw[k >> 1] >> 16 : mw[k >> 1] & 0xffffu);
            bool pos;
            if constexpr (MFY) pos = fmaf(yv.v[k], ms[k], mh[k]) > 0.f;
            else pos = m > 0;
            if (!pos) zw[k >> 1] &= (k & 1) ? 0x0000ffffu : 0xffff0000u;
            const float dz = pos ? gv.v[k] : 0.f;
            o.v[k] = ka[k] * (dz - db[k] - (yv.v[k] - mu[k]) * is[k] * dg[k]);
        }
        dzr = make_uint4(zw[0], zw[1], zw[2], zw[3]);
        return pack8(o);
    };
    size_t i = i0;
    for (; i + stride < total8; i += 2 * stride) {
        const uint4 g0 = ld16(g, i), g1 = ld16(g, i + stride);
        uint4 m0 = make_uint4(0, 0, 0, 0), m1 = m0;
        if constexpr (!MFY) { m0 = ld16(mask16, i); m1 = ld16(mask16, i + stride); }
        const uint4 y0 = ld16(y, i), y1 = ld16(y, i + stride);
        uint4 z0, z1;
        st16(dy, i, one(g0, m0, y0, z0));
        st16(dy, i + stride, one(g1, m1, y1, z1));
        if (dz_out) { st16(dz_out, i, z0); st16(dz_out, i + stride, z1); }
    }
    if (i < total8) {
        uint4 z0;
        uint4 mt = make_uint4(0, 0, 0, 0);
        if constexpr (!MFY) mt = ld16(mask16, i);
        st16(dy, i, one(ld16(g, i), mt, ld16(y, i), z0));
        if (dz_out) st16(dz_out, i, z0);
    }
}

// Head: a = relu(bn(y)) over the 32-channel map (BatchNorm 2, networks.py:24) and, in the same pass, z = conv3(a) WITHOUT bias
// (1x1, 32 -> Cout <= 4; its bias is added behind the second upsample, head.hip) -- the activation is not read back for the last layer,
// and forwards nothing is differentiated through (a == NULL) do not write it at all.  8 lanes share a pixel (one float4 of the 32
// channels each); z is NCHW [B][Cout][HW].
__global__ void __launch_bounds__(256) head_bn_relu_conv3_kernel(const float* __restrict__ y, BnRef bn, const float* __restrict__ w3,
                                                                  float* __restrict__ a, float* __restrict__ z, int B, int HW, int Cout) {
    constexpr int CIN = 32, L = CIN / 4, MAXC = 4;
    __shared__ float cs[2 * kCoeffMaxC];
    bn_coeff_block(bn, cs);
    __syncthreads();
    const int sub = threadIdx.x % L;
    const float4 sc = ld4(cs + sub * 4), sh = ld4(cs + kCoeffMaxC + sub * 4);
    float4 wv[MAXC];
#pragma unroll
    for (int co = 0; co < MAXC; ++co) wv[co] = co < Cout ? ld4(w3 + co * CIN + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned total = (unsigned)B * (unsigned)HW, ppb = 256 / L;
    for (unsigned pix = blockIdx.x * ppb + threadIdx.x / L; pix < total; pix += gridDim.x * ppb) {
        const float4 v = relu4(fma4(ld4(y + (size_t)pix * CIN + sub * 4), sc, sh));
        if (a) st4(a + (size_t)pix * CIN + sub * 4, v);
        const unsigned b = pix / (unsigned)HW, p = pix - b * (unsigned)HW;
#pragma unroll
        for (int co = 0; co < MAXC; ++co) {
            if (co < Cout) {
                float s = v.x * wv[co].x + v.y * wv[co].y + v.z * wv[co].z + v.w * wv[co].w;
#pragma unroll
                for (int o = L / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                if (sub == 0) z[((size_t)b * Cout + co) * HW + p] = s;
            }
        }
    }
    bn_commit(bn);
}

// stem: pooled = maxpool3x3 s2 p1 over relu(bn(y)); idx = first maximal window slot (dy*3+dx), scan order
__global__ void stem_pool_fwd_kernel(const float* __restrict__ y, BnRef bn, float* __restrict__ pooled,
                                     uint8_t* __restrict__ idx, Planes pl, int B, int H, int W, int C4, int y_bf16) {
    const int Ho = H / 2, Wo = W / 2;
    size_t total = (size_t)B * Ho * Wo * C4;
    // the grid stride (gridDim * 256) is a multiple of C4, so a thread keeps its 4 channels: BN coefficients once
    float4 sc, sh;
    bn_coeff4(bn, (int)((blockIdx.x * blockDim.x + threadIdx.x) % (unsigned)C4) * 4, sc, sh);
    // 32-bit index arithmetic (the launchers bound the element count): 64-bit div / mod cost ~10x as many instructions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        int c4 = (int)(i % (unsigned)C4);
        unsigned r = i / (unsigned)C4;
        int px = (int)(r % Wo); r /= Wo;
        int py = (int)(r % Ho);
        int b = (int)(r / Ho);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int iy = 2 * py - 1 + dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                int ix = 2 * px - 1 + dx;
                if (ix < 0 || ix >= W) continue;
                float4 v = relu4(fma4(ld4y(y, (((size_t)b * H + iy) * W + ix) * C4 + c4, y_bf16), sc, sh));
                unsigned char s = (unsigned char)(dy * 3 + dx);
                if (v.x > best.x) { best.x = v.x; bi.x = s; }
                if (v.y > best.y) { best.y = v.y; bi.y = s; }
                if (v.z > best.z) { best.z = v.z; bi.z = s; }
                if (v.w > best.w) { best.w = v.w; bi.w = s; }
            }
        }
        st4(pooled + i * 4, best);
        st_planes(pl, i, best);
        *reinterpret_cast<uchar4*>(idx + i * 4) = bi;
    }
    bn_commit(bn);
}

// gather form of maxpool backward fused with the ReLU mask: dz at HxW from g at (H/2)x(W/2).
// red != NULL: also accumulates the BatchNorm-backward sums of the layer in front (sum dz | sum dz * xhat, xhat from the saved pre-BN
// output y / mean / invstd) into one of `replicas` copies of a [2*C] slot (block index mod replicas; launch_stats_fold adds them up):
// dz is complete here, so the separate reduction pass over it is not needed
__global__ void __launch_bounds__(256) stem_pool_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pooled,
                                     const uint8_t* __restrict__ idx, float* __restrict__ dz, int B, int H, int W,
                                     int C4, int g_bf16, const float* __restrict__ y, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, double* red, int y_bf16, int replicas) {
    __shared__ double sm[4 * 16 * 8];                  // [wave][channel group <= 16][8]
    const int Ho = H / 2, Wo = W / 2;
    size_t total = (size_t)B * H * W * C4;
    // the grid stride (gridDim * 256) is a multiple of C4, so a thread keeps its 4 channels
    const int cfix = (int)((blockIdx.x * blockDim.x + threadIdx.x) % (unsigned)C4);
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, s0 = mu, s1 = mu;
    if (red) { mu = ld4(mean + cfix * 4); is = ld4(invstd + cfix * 4); }
    // 32-bit index arithmetic (the launchers bound the element count): 64-bit div / mod cost ~10x as many instructions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        int c4 = (int)(i % (unsigned)C4);
        unsigned r = i / (unsigned)C4;
        int x = (int)(r % W); r /= W;
        int yy = (int)(r % H);
        int b = (int)(r / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows (py, dy) with 2*py - 1 + dy == yy
        int py0 = yy >> 1, ny = (yy & 1) ? 2 : 1;
        int px0 = x >> 1, nx = (x & 1) ? 2 : 1;
        for (int a = 0; a < ny; ++a) {
            int py = py0 + a;
            if (py >= Ho) continue;
            int dy = yy - 2 * py + 1;
            for (int e = 0; e < nx; ++e) {
                int px = px0 + e;
                if (px >= Wo) continue;
                int dx = x - 2 * px + 1;
                unsigned char s = (unsigned char)(dy * 3 + dx);
                size_t o = ((((size_t)b * Ho + py) * Wo + px) * C4 + c4) * 4;
                uchar4 bi = *reinterpret_cast<const uchar4*>(idx + o);
                float4 pv = ld4(pooled + o), gv = ld4y(g, o / 4, g_bf16);
                if (bi.x == s && pv.x > 0.f) acc.x += gv.x;
                if (bi.y == s && pv.y > 0.f) acc.y += gv.y;
                if (bi.z == s && pv.z > 0.f) acc.z += gv.z;
                if (bi.w == s && pv.w > 0.f) acc.w += gv.w;
            }
        }
        st4(dz + i * 4, acc);
        if (red) {
            const float4 yv = ld4y(y, i, y_bf16);
            s0 = add4(s0, acc);
            s1.x += acc.x * ((yv.x - mu.x) * is.x); s1.y += acc.y * ((yv.y - mu.y) * is.y);
            s1.z += acc.z * ((yv.z - mu.z) * is.z); s1.w += acc.w * ((yv.w - mu.w) * is.w);
        }
    }
    if (!red) return;
    // lanes l, l + C4, ... of a wave hold the same channels: fp64 butterfly, one LDS slot per wave and channel group, C4 threads push
    double v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    for (int off = C4; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += __shfl_xor(v[k], off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (lane < C4)
#pragma unroll
        for (int k = 0; k < 8; ++k) sm[(wave * C4 + lane) * 8 + k] = v[k];
    __syncthreads();
    if (tid < C4) {
        const int C = C4 * 4;
        red += (size_t)(blockIdx.x % (unsigned)replicas) * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(red + tid * 4 + k, sm[tid * 8 + k] + sm[(C4 + tid) * 8 + k] + sm[(2 * C4 + tid) * 8 + k] + sm[(3 * C4 + tid) * 8 + k]);
            unsafeAtomicAdd(red + C + tid * 4 + k, sm[tid * 8 + 4 + k] + sm[(C4 + tid) * 8 + 4 + k] + sm[(2 * C4 + tid) * 8 + 4 + k] + sm[(3 * C4 + tid) * 8 + 4 + k]);
        }
    }
}

// per-channel reductions over rows: threads = [rowlanes][C4]; block partials -> fp64 atomics
template <int MODE>   // 0: bn backward (sum dz, sum dz*xhat)   1: plain column sum
__global__ void __launch_bounds__(256) chan_reduce_kernel(const float* __restrict__ g, const float* __restrict__ mask,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, double* red, size_t rows,
                                                          int C4, int y_bf16, int g_bf16, int replicas) {
    __shared__ double sm[256 * 8];
    const int tid = threadIdx.x;
    const int rowlanes = 256 / C4;
    const int c4 = tid % C4, rl = tid / C4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < rowlanes) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
        if (MODE == 0) { mu = ld4(mean + c4 * 4); is = ld4(invstd + c4 * 4); }
        for (size_t r = (size_t)blockIdx.x * rowlanes + rl; r < rows; r += (size_t)gridDim.x * rowlanes) {
            size_t o = (r * C4 + c4) * 4;
            float4 dz = ld4y(g, o / 4, g_bf16);
            if (MODE == 0) {
                if (mask) {
                    float4 m = ld4(mask + o);
                    dz.x = m.x > 0.f ? dz.x : 0.f; dz.y = m.y > 0.f ? dz.y : 0.f;
                    dz.z = m.z > 0.f ? dz.z : 0.f; dz.w = m.w > 0.f ? dz.w : 0.f;
                }
                float4 yv = ld4y(y, o / 4, y_bf16);
                t.x += dz.x * ((yv.x - mu.x) * is.x); t.y += dz.y * ((yv.y - mu.y) * is.y);
                t.z += dz.z * ((yv.z - mu.z) * is.z); t.w += dz.w * ((yv.w - mu.w) * is.w);
            }
            s = add4(s, dz);
        }
    }
    double* my = sm + tid * 8;
    my[0] = s.x; my[1] = s.y; my[2] = s.z; my[3] = s.w;
    my[4] = t.x; my[5] = t.y; my[6] = t.z; my[7] = t.w;
    __syncthreads();
    if (tid < C4) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = 0; l < rowlanes; ++l)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += sm[(l * C4 + tid) * 8 + k];
        const int C = C4 * 4;
        // all blocks finish together and their fp64 atomics into the same few cache lines serialise (~6 ns per atomic and line: 288
        // blocks x 16 addresses per line spent 26 of their 35 us queueing):
        // block b adds into copy b % replicas of the slot, the finishing launch adds the copies
        red += (size_t)(blockIdx.x % (unsigned)replicas) * (MODE == 0 ? 2 * C : C);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(red + tid * 4 + k, a[k]);
            if (MODE == 0) unsafeAtomicAdd(red + C + tid * 4 + k, a[4 + k]);
        }
    }
}

// mask / mask_pl: the ReLU output whose sign gates g, as fp32 or as its bf16 (hi) plane; dy may be NULL (planes only)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ mask, const uint16_t* __restrict__ mask16,
                                    const float* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const double* __restrict__ red, float* __restrict__ dy,
                                    float* __restrict__ dz_out, float* dgamma, float* dbeta, Planes pl, size_t rows, int C4,
                                    int y_bf16, float inv_rows, float dparam_scale, int g_bf16, const float* __restrict__ mscale,
                                    const float* __restrict__ mshift) {
    const int C = C4 * 4;
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dbeta[c] = (float)red[c] * dparam_scale;
            dgamma[c] = (float)red[C + c] * dparam_scale;
        }
    }
    size_t total4 = rows * C4;
    // the grid stride is a multiple of C4 (256 % C4 == 0): a thread keeps its 4 channels
    const int c = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) % C4) * 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        float4 dz = ld4y(g, i, g_bf16);
        const float4 yv = ld4y(y, i, y_bf16);
        if (mask) {
            float4 m = ld4(mask + i * 4);
            dz.x = m.x > 0.f ? dz.x : 0.f; dz.y = m.y > 0.f ? dz.y : 0.f;
            dz.z = m.z > 0.f ? dz.z : 0.f; dz.w = m.w > 0.f ? dz.w : 0.f;
        } else if (mask16) {
            const ushort4 m = *reinterpret_cast<const ushort4*>(mask16 + i * 4);
            dz.x = (short)m.x > 0 ? dz.x : 0.f; dz.y = (short)m.y > 0 ? dz.y : 0.f;
            dz.z = (short)m.z > 0 ? dz.z : 0.f; dz.w = (short)m.w > 0 ? dz.w : 0.f;
        } else if (mscale) {                 // the activation was never stored: its sign from the pre-BN output (common.h InBn)
            const float4 a = fma4(yv, ld4(mscale + c), ld4(mshift + c));
            dz.x = a.x > 0.f ? dz.x : 0.f; dz.y = a.y > 0.f ? dz.y : 0.f;
            dz.z = a.z > 0.f ? dz.z : 0.f; dz.w = a.w > 0.f ? dz.w : 0.f;
        }
        if (dz_out) {
            if (g_bf16) *reinterpret_cast<ushort4*>(reinterpret_cast<uint16_t*>(dz_out) + i * 4) = make_ushort4((uint16_t)(__float_as_uint(dz.x) >> 16), (uint16_t)(__float_as_uint(dz.y) >> 16), (uint16_t)(__float_as_uint(dz.z) >> 16), (uint16_t)(__float_as_uint(dz.w) >> 16));   // (exact: dz is a bf16 value or 0)
            else st4(dz_out + i * 4, dz);
        }
        const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4(gamma + c);
        float db[4] = {(float)red[c], (float)red[c + 1], (float)red[c + 2], (float)red[c + 3]};
        float dg[4] = {(float)red[C + c], (float)red[C + c + 1], (float)red[C + c + 2], (float)red[C + c + 3]};
        float4 o;
        o.x = ga.x * is.x * (dz.x - db[0] * inv_rows - (yv.x - mu.x) * is.x * dg[0] * inv_rows);
        o.y = ga.y * is.y * (dz.y - db[1] * inv_rows - (yv.y - mu.y) * is.y * dg[1] * inv_rows);
        o.z = ga.z * is.z * (dz.z - db[2] * inv_rows - (yv.z - mu.z) * is.z * dg[2] * inv_rows);
        o.w = ga.w * is.w * (dz.w - db[3] * inv_rows - (yv.w - mu.w) * is.w * dg[3] * inv_rows);
        if (dy) st4(dy + i * 4, o);
        st_planes(pl, i, o);
    }
}

// out[i] = sum over the replicas of rep[r][i]   (n = 2 * C entries)
__global__ void stats_fold_kernel(const double* __restrict__ rep, double* __restrict__ out, int n, int replicas) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int r = 0; r < replicas; ++r) a += rep[(size_t)r * n + i];
    out[i] = a;
}

__global__ void colsum_finish_kernel(const double* __restrict__ red, float* out, int C, int replicas) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0;
    for (int r = 0; r < replicas; ++r) a += red[(size_t)r * C + c];
    out[c] = (float)a;
}

// bilinear x2, align_corners=True (ATen upsample_bilinear2d semantics: float source index, lambda clamp)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_coord(int o, int in_size, float scale) {
    float real = scale * (float)o;
    int i0 = (int)real;
    if (i0 > in_size - 1) i0 = in_size - 1;
    int off = (i0 < in_size - 1) ? 1 : 0;
    float l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
    Lerp r;
    r.i0 = i0; r.i1 = i0 + off; r.l1 = l1; r.l0 = 1.f - l1;
    return r;
}

// red != NULL: also accumulates the per-channel sum / sum of squares of the outputs (the train-mode BatchNorm statistics of a layer
// whose convolution ran BEFORE the upsample, forward.hip head); relu: rectify the outputs
__global__ void __launch_bounds__(256) upsample2x_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, Planes pl, int B, int H, int W,
                                      int C4, double* red, int relu, int replicas) {
    __shared__ double sm[4 * 16 * 8];                 // [wave][channel group <= 16][8]
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    size_t total = (size_t)B * Ho * Wo * C4;
    // 32-bit index arithmetic (the launchers bound the element count): 64-bit div / mod cost ~10x as many instructions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        int c4 = (int)(i % (unsigned)C4);
        unsigned r = i / (unsigned)C4;
        int ox = (int)(r % Wo); r /= Wo;
        int oy = (int)(r % Ho);
        int b = (int)(r / Ho);
        Lerp ly = lerp_coord(oy, H, sy), lx = lerp_coord(ox, W, sx);
        const float* base = in + (size_t)b * H * W * C4 * 4 + c4 * 4;
        float4 v00 = ld4(base + ((size_t)ly.i0 * W + lx.i0) * C4 * 4), v01 = ld4(base + ((size_t)ly.i0 * W + lx.i1) * C4 * 4);
        float4 v10 = ld4(base + ((size_t)ly.i1 * W + lx.i0) * C4 * 4), v11 = ld4(base + ((size_t)ly.i1 * W + lx.i1) * C4 * 4);
        float4 o;
        o.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
        o.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
        o.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
        o.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
        if (relu) o = relu4(o);
        if (out) st4(out + i * 4, o);
        st_planes(pl, i, o);
        if (red) { s0 = add4(s0, o); s1.x += o.x * o.x; s1.y += o.y * o.y; s1.z += o.z * o.z; s1.w += o.w * o.w; }
    }
    if (!red) return;
    // (the grid stride is a multiple of C4 -- the launcher checks 64 % C4 == 0 --, so a thread kept its 4 channels.)  Lanes l, l + C4,
    // l + 2 C4 ... of a wave hold the same channels: butterfly over them in fp64, one LDS slot per wave and channel group, then C4
    // threads push the block's sums.  (The first version summed 256 fp64 partials per channel group serially in LDS: 57 us per launch.)
    double v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    for (int off = C4; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += __shfl_xor(v[k], off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (lane < C4)
#pragma unroll
        for (int k = 0; k < 8; ++k) sm[(wave * C4 + lane) * 8 + k] = v[k];
    __syncthreads();
    if (tid < C4) {
        const int C = C4 * 4;
        red += (size_t)(blockIdx.x % (unsigned)replicas) * 2 * C;     // (same-address atomics of blocks that finish together serialise)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsafeAtomicAdd(red + tid * 4 + k, sm[tid * 8 + k] + sm[(C4 + tid) * 8 + k] + sm[(2 * C4 + tid) * 8 + k] + sm[(3 * C4 + tid) * 8 + k]);
            unsafeAtomicAdd(red + C + tid * 4 + k, sm[tid * 8 + 4 + k] + sm[(C4 + tid) * 8 + 4 + k] + sm[(2 * C4 + tid) * 8 + 4 + k] + sm[(3 * C4 + tid) * 8 + 4 + k]);
        }
    }
}

// transpose of the above as a gather: din[iy][ix] = sum over outputs that interpolate from (iy, ix)
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int H, int W,
                                      int C4, Planes pl) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    size_t total = (size_t)B * H * W * C4;
    // 32-bit index arithmetic (the launchers bound the element count): 64-bit div / mod cost ~10x as many instructions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        int c4 = (int)(i % (unsigned)C4);
        unsigned r = i / (unsigned)C4;
        int ix = (int)(r % W); r /= W;
        int iy = (int)(r % H);
        int b = (int)(r / H);
        // candidate outputs: source index in (iy-1, iy+1)  ->  o in ((iy-1)/s, (iy+1)/s), padded by 1
        int oy_lo = max(0, (int)floorf((float)(iy - 1) / sy) - 1), oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) / sy) + 1);
        int ox_lo = max(0, (int)floorf((float)(ix - 1) / sx) - 1), ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) / sx) + 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            Lerp ly = lerp_coord(oy, H, sy);
            float wy = (ly.i0 == iy ? ly.l0 : 0.f) + (ly.i1 == iy ? ly.l1 : 0.f);
            if (ly.i0 == ly.i1 && ly.i0 == iy) wy = ly.l0 + ly.l1;
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                Lerp lx = lerp_coord(ox, W, sx);
                float wx = (lx.i0 == ix ? lx.l0 : 0.f) + (lx.i1 == ix ? lx.l1 : 0.f);
                if (wx == 0.f) continue;
                float4 gv = ld4(dout + ((((size_t)b * Ho + oy) * Wo + ox) * C4 + c4) * 4);
                float w = wy * wx;
                acc.x += w * gv.x; acc.y += w * gv.y; acc.z += w * gv.z; acc.w += w * gv.w;
            }
        }
        st4(din + i * 4, acc);
        st_planes(pl, i, acc);
    }
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        st4(dst + i * 4, add4(ld4(dst + i * 4), ld4(src + i * 4)));
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / HW, p = i - b * HW;
        for (int c = 0; c < C; ++c) out[i * C + c] = in[(b * C + c) * HW + p];
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW) {
    size_t total = (size_t)B * HW;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / HW, p = i - b * HW;
        for (int c = 0; c < C; ++c) out[(b * C + c) * HW + p] = in[i * C + c];
    }
}

inline int grid_for(size_t work_items, int block = 256, int cap = 256 * 8) {
    size_t b = (work_items + block - 1) / block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

namespace {
// running-statistics update of a train-mode BatchNorm from its (globally reduced) [sum | sum of squares] alone -- the commit
// bn_commit performs inside the consuming kernel, for a rank that has no rows of its own to normalise (simq_forward_sync_null)
__global__ void bn_running_update_kernel(const double* __restrict__ stats, float* __restrict__ rmean, float* __restrict__ rvar,
                                         double rows, int C) {
    const double inv_rows = 1.0 / rows;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double mean = stats[c] * inv_rows;
        double var = stats[C + c] * inv_rows - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        rmean[c] = (float)(BN_MOMENTUM * mean + (1.0 - BN_MOMENTUM) * (double)rmean[c]);
        rvar[c] = (float)(BN_MOMENTUM * unbiased + (1.0 - BN_MOMENTUM) * (double)rvar[c]);
    }
}
}  // namespace

int launch_bn_running_update(const double* stats, float* rmean, float* rvar, double rows, int C, hipStream_t stream) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(1), dim3(256), 0, stream, stats, rmean, rvar, rows, C);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_bn_eval_coeff(const BnEvalTable& t, const float* params, const float* bnbuf, float* aux, hipStream_t stream) {
    hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3(t.n), dim3(256), 0, stream, t, params, bnbuf, aux);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_bn_apply(const float* y, const BnRef& bn, const float* res, const BnRef* rbn, int relu, float* out, int64_t rows,
                    int C, hipStream_t stream, Planes pl, Planes res_pl, int y_bf16) {
    SIMQ_REQUIRE(C % 4 == 0 && 256 % (C / 4) == 0, "bn_apply: C=%d unsupported", C);
    SIMQ_REQUIRE(out || pl.hi, "bn_apply: no output requested");
    if (y_bf16 && !out && pl.hi && !pl.lo && !res && !res_pl.lo && C % 8 == 0 && 256 % (C / 8) == 0 && C <= 512) {   // all-bf16 form
        size_t total8 = (size_t)rows * (C / 8);
        note_launch("bn_apply16");
        hipLaunchKernelGGL(bn_apply16_kernel, dim3(grid_for(total8)), dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(y), bn, res_pl.hi,
                           rbn ? *rbn : bn, rbn ? 1 : 0, relu, pl.hi, total8, C / 8);
        SIMQ_CHECK_LAUNCH();
        return 0;
    }
    size_t total4 = (size_t)rows * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, y, bn, res, res_pl, rbn ? *rbn : bn,
                       rbn ? 1 : 0, relu, out, pl, total4, C / 4, y_bf16);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_head_bn_relu_conv3(const float* y, const BnRef& bn, const float* w3, float* a, float* z, int B, int HW, int Cout, hipStream_t stream) {
    SIMQ_REQUIRE(bn.C == 32 && Cout >= 1 && Cout <= 4, "head_bn_relu_conv3: C=%d Cout=%d unsupported", bn.C, Cout);
    SIMQ_REQUIRE((size_t)B * HW < 2147483648ull, "head_bn_relu_conv3: too many pixels for 32-bit indexing");
    size_t blocks = ((size_t)B * HW + 31) / 32;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(head_bn_relu_conv3_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, bn, w3, a, z, B, HW, Cout);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_stem_pool_fwd(const float* y, const BnRef& bn, float* pooled, uint8_t* idx, int B, int H, int W, int C,
                         hipStream_t stream, Planes pl, int y_bf16) {
    SIMQ_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && 256 % (C / 4) == 0, "stem_pool: bad shape");
    size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    SIMQ_REQUIRE(total < 2147483648ull, "stem_pool: tensor too large for 32-bit indexing");
    hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, y, bn, pooled, idx, pl, B, H,
                       W, C / 4, y_bf16);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_stem_pool_bwd(const float* g, const float* pooled, const uint8_t* idx, float* dz, int B, int H, int W,
                         int C, hipStream_t stream, int g_bf16, const float* y, const float* mean, const float* invstd, double* red,
                         int y_bf16, int replicas) {
    size_t total = (size_t)B * H * W * (C / 4);
    SIMQ_REQUIRE(total < 2147483648ull, "stem_pool_bwd: tensor too large for 32-bit indexing");
    SIMQ_REQUIRE(!red || (y && mean && invstd && C % 4 == 0 && C / 4 <= 16 && 64 % (C / 4) == 0 && replicas >= 1),
                 "stem_pool_bwd: fused reduction needs y / mean / invstd and C/4 in {1,2,4,8,16}");
    hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(grid_for(total, 256, red ? 1024 : 2048)), dim3(256), 0, stream, g, pooled, idx, dz, B, H, W,
                       C / 4, g_bf16, y, mean, invstd, red, y_bf16, replicas > 0 ? replicas : 1);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

static int reduce_grid(int64_t rows, int C4, int rows_per_lane = 32) {
    int rowlanes = 256 / C4;
    int64_t blocks = (rows + (int64_t)rowlanes * rows_per_lane - 1) / ((int64_t)rowlanes * rows_per_lane);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int launch_bn_bwd_reduce(const float* g, const float* mask, const float* y, const float* mean, const float* invstd,
                         double* red, int64_t rows, int C, hipStream_t stream, int y_bf16, int g_bf16, int replicas) {
    SIMQ_REQUIRE(C % 4 == 0 && C / 4 <= 256, "bn_bwd_reduce: C=%d unsupported", C);
    // replicas > 1: `red` is a zeroed scratch of replicas * 2C doubles (the caller folds it), four times the blocks
    hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(reduce_grid(rows, C / 4, replicas > 1 ? 8 : 32)), dim3(256), 0, stream, g, mask, y, mean,
                       invstd, red, (size_t)rows, C / 4, y_bf16, g_bf16, replicas > 1 ? replicas : 1);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_bn_bwd_apply(const float* g, const float* mask, const float* y, const float* mean, const float* invstd,
                        const float* gamma, const double* red, float* dy, float* dz_out, float* dgamma, float* dbeta,
                        int64_t rows, int C, hipStream_t stream, Planes pl, const uint16_t* mask16, int y_bf16, double global_rows,
                        float dparam_scale, int g_bf16, const float* mscale, const float* mshift) {
    SIMQ_REQUIRE(dy || pl.hi, "bn_bwd_apply: no output requested");
    SIMQ_REQUIRE(C % 4 == 0 && 256 % (C / 4) == 0, "bn_bwd_apply: C=%d unsupported", C);
    SIMQ_REQUIRE(!mscale || (mshift && !mask && !mask16), "bn_bwd_apply: the recomputed mask (mscale / mshift) excludes a mask tensor");
    if (g_bf16 && y_bf16 && (mask16 || mscale) && !mask && !dy && pl.hi && !pl.lo && C % 8 == 0 && 256 % (C / 8) == 0) {   // all-bf16 form
        size_t total8 = (size_t)rows * (C / 8);
        const float inv_rows = (float)(1.0 / (global_rows > 0.0 ? global_rows : (double)rows));
        note_launch(mscale ? "bn_bwd_apply16_mask_from_y" : "bn_bwd_apply16");
        if (mscale) hipLaunchKernelGGL(bn_bwd_apply16_kernel<true>, dim3(grid_for(total8)), dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(g), mask16,
                                       reinterpret_cast<const uint16_t*>(y), mean, invstd, gamma, red, pl.hi, reinterpret_cast<uint16_t*>(dz_out), dgamma, dbeta,
                                       total8, C / 8, inv_rows, dparam_scale, mscale, mshift);
        else hipLaunchKernelGGL(bn_bwd_apply16_kernel<false>, dim3(grid_for(total8)), dim3(256), 0, stream, reinterpret_cast<const uint16_t*>(g), mask16,
                                reinterpret_cast<const uint16_t*>(y), mean, invstd, gamma, red, pl.hi, reinterpret_cast<uint16_t*>(dz_out), dgamma, dbeta,
                                total8, C / 8, inv_rows, dparam_scale, mscale, mshift);
        SIMQ_CHECK_LAUNCH();
        return 0;
    }
    size_t total4 = (size_t)rows * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, stream, g, mask, mask16, y, mean, invstd,
                       gamma, red, dy, dz_out, dgamma, dbeta, pl, (size_t)rows, C / 4, y_bf16,
                       (float)(1.0 / (global_rows > 0.0 ? global_rows : (double)rows)), dparam_scale, g_bf16, mscale, mshift);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_colsum(const float* x, double* red_scratch, float* out, int64_t rows, int C, hipStream_t stream) {
    SIMQ_REQUIRE(C % 4 == 0 && C / 4 <= 256, "colsum: C=%d unsupported", C);
    SIMQ_CHECK_HIP(hipMemsetAsync(red_scratch, 0, sizeof(double) * C, stream));
    hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(reduce_grid(rows, C / 4)), dim3(256), 0, stream, x, nullptr, nullptr,
                       nullptr, nullptr, red_scratch, (size_t)rows, C / 4, 0, 0, 1);
    SIMQ_CHECK_LAUNCH();
    return launch_colsum_finish(red_scratch, out, C, stream);
}

// the same with `replicas` copies of the scratch slot (red_scratch: replicas * C doubles) and four times the blocks
int launch_colsum_rep(const float* x, double* red_scratch, float* out, int64_t rows, int C, int replicas, hipStream_t stream) {
    SIMQ_REQUIRE(C % 4 == 0 && C / 4 <= 256 && replicas >= 1, "colsum: C=%d unsupported", C);
    SIMQ_CHECK_HIP(hipMemsetAsync(red_scratch, 0, sizeof(double) * C * replicas, stream));
    hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(reduce_grid(rows, C / 4, 8)), dim3(256), 0, stream, x, nullptr, nullptr,
                       nullptr, nullptr, red_scratch, (size_t)rows, C / 4, 0, 0, replicas);
    SIMQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, red_scratch, out, C, replicas);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

__global__ void bn_running_deferred_kernel(float* __restrict__ buf, const double* __restrict__ defer, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = (float)(BN_MOMENTUM * defer[i] + (1.0 - BN_MOMENTUM) * (double)buf[i]);
}

int launch_bn_running_deferred(float* bnbuf, const double* defer, int64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(bn_running_deferred_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, bnbuf, defer, n);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_stats_fold(const double* rep, double* out, int n, int replicas, hipStream_t stream) {
    hipLaunchKernelGGL(stats_fold_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, rep, out, n, replicas);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_colsum_finish(const double* red, float* out, int C, hipStream_t stream) {
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, red, out, C, 1);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_upsample2x_fwd(const float* in, float* out, int B, int H, int W, int C, hipStream_t stream, Planes pl, double* stats, int relu,
                          int replicas) {
    SIMQ_REQUIRE(C % 4 == 0, "upsample: C=%d must be a multiple of 4", C);
    SIMQ_REQUIRE(!stats || (C / 4 <= 16 && 64 % (C / 4) == 0 && !relu), "upsample2x_fwd: fused statistics need C/4 in {1,2,4,8,16} (C=%d) and no ReLU", C);
    size_t total = (size_t)B * 4 * H * W * (C / 4);
    SIMQ_REQUIRE(total < 2147483648ull, "upsample2x_fwd: tensor too large for 32-bit indexing");
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for(total, 256, stats ? 1024 : 2048)), dim3(256), 0, stream, in, out, pl, B, H, W, C / 4, stats, relu,
                       replicas > 0 ? replicas : 1);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_upsample2x_bwd(const float* dout, float* din, int B, int H, int W, int C, hipStream_t stream, Planes pl) {
    SIMQ_REQUIRE(C % 4 == 0, "upsample: C=%d must be a multiple of 4", C);
    size_t total = (size_t)B * H * W * (C / 4);
    SIMQ_REQUIRE(total < 2147483648ull, "upsample2x_bwd: tensor too large for 32-bit indexing");
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dout, din, B, H, W, C / 4, pl);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_add_inplace(float* dst, const float* src, int64_t n, hipStream_t stream) {
    SIMQ_REQUIRE(n % 4 == 0, "add_inplace: n must be a multiple of 4");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for((size_t)n / 4)), dim3(256), 0, stream, dst, src, (size_t)n / 4);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, hipStream_t stream) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)B * HW)), dim3(256), 0, stream, in, out, B, C, HW);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, hipStream_t stream) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)B * HW)), dim3(256), 0, stream, in, out, B, C, HW);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
