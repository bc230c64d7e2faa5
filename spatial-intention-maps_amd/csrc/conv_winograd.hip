// Winograd F(2x2, 3x3) convolution in fp32 for the wide 3x3 / stride-1 / pad-1 layers (layer4 of the encoder: 73 % of the
// network's multiply-adds).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A          per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplies per 2x2 outputs instead of 36: the 3x3 convolution becomes 16 independent GEMMs
//   Mt[g][t][co] = sum_ci V[g][t][ci] * U[g][co][ci]        t = (b, tile_y, tile_x), T = B * (H/2) * (W/2)
// with 2.25x fewer matrix-core FLOPs than the implicit GEMM of conv_igemm.hip.  All arithmetic stays fp32 (the transform
// matrices only hold 0, +-1, +-1/2), the result differs from the direct convolution by a few fp32 ulps of the accumulated
// magnitude (tests/test_gpu_ops.py: <= 2e-6 of the output range, the bar for the network outputs is 1e-4).
//
// Three launches per convolution (the GEMM is the batched form of igemm_conv_kernel, conv_igemm.hip):
//   wino_input_kernel    x [B][H][W][Cin] (NHWC)        -> V  [16][T][Cin]      HBM-bound: reads x (4x from L2), writes 4x |x|
//   launch_gemm_batched  V, U [16][Cout][Cin]           -> Mt [16][T][Cout]     MFMA-bound
//   wino_output_kernel   Mt                             -> y  [B][H][W][Cout]   HBM-bound, carries the WHOLE conv epilogue:
//        +bias, BatchNorm batch statistics, folded-BN affine, residual add, ReLU, fused BN-backward reduction (dgrad) --
//        the same operations in the same order as igemm_epilogue.h.
// U = G g G^T is part of the weight cache (wino_weight_kernel, refreshed by simq_weights_prepare with the other derived
// weights); a dgrad is the same convolution over the flipped / transposed weight.
#include "common.h"
#include "bn_coeff.h"
#include "igemm_epilogue.h"

namespace simq {

namespace {

// ---- U[g][co][ci] = (G w[co][:, :][ci] G^T)[g / 4][g % 4],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] ----
__global__ void __launch_bounds__(256) wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int cout, int cin) {
    const int total = cout * cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i / cin, ci = i - co * cin;
        const float* g = w + (size_t)co * 9 * cin + ci;
        float t[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = g[(0 * 3 + c) * cin], g1 = g[(1 * 3 + c) * cin], g2 = g[(2 * 3 + c) * cin];
            t[0][c] = g0;
            t[1][c] = 0.5f * (g0 + g1 + g2);
            t[2][c] = 0.5f * (g0 - g1 + g2);
            t[3][c] = g2;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
            U[(size_t)(r * 4 + 0) * total + i] = u0;
            U[(size_t)(r * 4 + 1) * total + i] = u1;
            U[(size_t)(r * 4 + 2) * total + i] = u2;
            U[(size_t)(r * 4 + 3) * total + i] = u3;
        }
    }
}

// ---- F(4x4, 3x3) for the NO-GRAD forwards (target net, double-DQN argmax forward) ------------------------------------------
// 36 products per 4x4 outputs: 1.78x fewer matrix FLOPs than F(2x2,3x3) and 2.25x instead of 4x transform volume.  Interpolation
// points {0, 1, -1, 1/2, -2, inf} (Cook-Toom; the symmetric textbook set {0, +-1, +-2} loses 2.4x more accuracy): measured / simulated
// fp32 error 3.6e-6 of the output range per layer against 6e-7 for F(2x2,3x3) -- ~1e-5 on the Q-map after the eight wide layers,
// a tenth of the 1e-4 parity bar.  First used only where NOTHING is differentiated through the result (the forwards that produce
// the TD target's bootstrap value and the greedy next action, train.py:119-124); since then also by the dgrads and by the grad-mode
// forward of layer4's 512->512 convolutions (simq_plan_options.winograd_f4_grad / winograd_f4_fwd_grad_min_cc, further down) --
// the rest of the grad-mode forward stays on F(2x2,3x3): its round-off is the gradient's error.
//   A^T = [[1,1,1,1,1,0],[0,1,-1,1/2,-2,0],[0,1,1,1/4,4,0],[0,1,-1,1/8,-8,1]]
//   G   = [[1,0,0],[1/3,1/3,1/3],[-1/3,1/3,-1/3],[-16/15,-8/15,-4/15],[1/15,-2/15,4/15],[0,0,1]]
//   B^T = [[1,-3/2,-2,3/2,1,0],[0,-1,1/2,5/2,1,0],[0,1,-5/2,1/2,1,0],[0,-2,-1,2,1,0],[0,1/2,-1,-1/2,1,0],[0,1,-3/2,-2,3/2,1]]
__device__ __forceinline__ void w4f_g(const float (&v)[3], float (&r)[6]) {     // r = G v
    r[0] = v[0];
    r[1] = (1.f / 3.f) * (v[0] + v[1] + v[2]);
    r[2] = (1.f / 3.f) * (v[1] - v[0] - v[2]);
    r[3] = -(4.f / 15.f) * (4.f * v[0] + 2.f * v[1] + v[2]);
    r[4] = (1.f / 15.f) * (v[0] - 2.f * v[1] + 4.f * v[2]);
    r[5] = v[2];
}
__device__ __forceinline__ void w4f_bt(floatx4 (&a)[6]) {                       // a <- B^T a
    const floatx4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    a[0] = a0 + 1.5f * (a3 - a1) - 2.f * a2 + a4;
    a[1] = 0.5f * a2 - a1 + 2.5f * a3 + a4;
    a[2] = a1 - 2.5f * a2 + 0.5f * a3 + a4;
    a[3] = 2.f * (a3 - a1) - a2 + a4;
    a[4] = 0.5f * (a1 - a3) - a2 + a4;
    a[5] = a1 - 1.5f * a2 - 2.f * a3 + 1.5f * a4 + a5;
}
__device__ __forceinline__ void w4f_at(const floatx4 (&m)[6], floatx4 (&y)[4]) { // y = A^T m
    const floatx4 s = m[1] + m[2], d = m[1] - m[2];
    y[0] = m[0] + s + m[3] + m[4];
    y[1] = d + 0.5f * m[3] - 2.f * m[4];
    y[2] = s + 0.25f * m[3] + 4.f * m[4];
    y[3] = d + 0.125f * m[3] - 8.f * m[4] + m[5];
}

// operand load of the input transforms (XBN): the saved pre-BatchNorm output with the BatchNorm + ReLU of the layer in between applied on
// the way in (common.h InBn: the fma / max sequence of bn_apply, so the transform sees the values bn_apply would have stored).  The load
// itself is UNCONDITIONAL from a clamped in-image address and the zero padding a select behind it: every load of a tile can be in flight
// before the first one is consumed (a load under a branch is waited for inside its branch).
template <bool XBN>
__device__ __forceinline__ floatx4 ld_tap(const float* __restrict__ img, int yy, int xx, int H, int W, int C, const floatx4& sc, const floatx4& sh) {
    const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    if constexpr (XBN) {
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        const floatx4 v = inbn_apply(*reinterpret_cast<const floatx4*>(img + ((size_t)yc * W + xc) * C), sc, sh);
        return ok ? v : floatx4{0.f, 0.f, 0.f, 0.f};
    } else {
        if (ok) return *reinterpret_cast<const floatx4*>(img + ((size_t)yy * W + xx) * C);
        return floatx4{0.f, 0.f, 0.f, 0.f};
    }
}

// every eligible convolution's U in one launch: blockIdx.y = table entry (source: the OHWI weight in the parameter buffer, or
// its flipped / transposed dgrad form in the weight cache)
__global__ void __launch_bounds__(256) wino_weight_all_kernel(const float* __restrict__ params, const float* __restrict__ wt,
                                                              float* __restrict__ ubase, const WinoWeightTable tab) {
    const WinoWeightDesc d = tab.d[blockIdx.y];
    const float* w = (d.from_wt ? wt : params) + d.src_off;
    float* U = ubase + d.u_off;
    const int cin = d.cin, total = d.cout * d.cin;
    if (d.pad_ == 1) {                                  // F(4x4,3x3) forward form: U4[r * 6 + c] = (G4 w G4^T)[r][c], 36 planes
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const int co = i / cin, ci = i - co * cin;
            const float* g = w + (size_t)co * 9 * cin + ci;
            float t[6][3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v[3] = {g[(0 * 3 + c) * cin], g[(1 * 3 + c) * cin], g[(2 * 3 + c) * cin]};
                float r[6];
                w4f_g(v, r);
#pragma unroll
                for (int k = 0; k < 6; ++k) t[k][c] = r[k];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float u[6];
                w4f_g(t[r], u);
#pragma unroll
                for (int c = 0; c < 6; ++c) U[(size_t)(r * 6 + c) * total + i] = u[c];
            }
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i / cin, ci = i - co * cin;
        const float* g = w + (size_t)co * 9 * cin + ci;
        float t[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = g[(0 * 3 + c) * cin], g1 = g[(1 * 3 + c) * cin], g2 = g[(2 * 3 + c) * cin];
            t[0][c] = g0;
            t[1][c] = 0.5f * (g0 + g1 + g2);
            t[2][c] = 0.5f * (g0 - g1 + g2);
            t[3][c] = g2;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            U[(size_t)(r * 4 + 0) * total + i] = t[r][0];
            U[(size_t)(r * 4 + 1) * total + i] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
            U[(size_t)(r * 4 + 2) * total + i] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
            U[(size_t)(r * 4 + 3) * total + i] = t[r][2];
        }
    }
}

// ---- V[g][t][c] = (B^T d B)[g / 4][g % 4],  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] ----
// One thread: one tile x 4 channels (float4); the C/4 lanes of a tile read / write full contiguous rows of C floats.
template <bool XBN>
__global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C,
                                                         int T, const InBn in) {
    const int lanes = C >> 2;                       // float4 lanes per tile
    const int tpb = 256 / lanes;                    // tiles per block iteration
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes;
    const int th = H >> 1, tw = W >> 1;
    const size_t gstride = (size_t)T * C;
    floatx4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XBN) inbn_coeff4(in, cl * 4, bsc, bsh);
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
        const float* img = x + (size_t)b * H * W * C + cl * 4;
        floatx4 d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = ld_tap<XBN>(img, y0 + i, x0 + j, H, W, C, bsc, bsh);
        floatx4 e[4][4];                            // B^T d
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[0][j] = d[0][j] - d[2][j];
            e[1][j] = d[1][j] + d[2][j];
            e[2][j] = d[2][j] - d[1][j];
            e[3][j] = d[1][j] - d[3][j];
        }
        float* dst = V + (size_t)t * C + cl * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 0) * gstride) = e[i][0] - e[i][2];
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 1) * gstride) = e[i][1] + e[i][2];
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 2) * gstride) = e[i][2] - e[i][1];
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 3) * gstride) = e[i][1] - e[i][3];
        }
    }
    if constexpr (XBN) inbn_commit(in);
}

// ---- y = A^T m A (+ epilogue),  A^T = [[1,1,1,0],[0,1,-1,-1]] ----
// One thread: one tile x 4 channels; a block walks its share of the tiles and keeps the per-channel partial sums of the
// statistics in registers (fp32 over <= a few dozen values), then reduces them across its tile lanes through LDS and adds
// them to the fp64 accumulators -- one atomic per channel and block, like the implicit-GEMM epilogue.
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ Mt, const EpiArgs p, int B, int H, int W, int C, int T) {
    __shared__ float red[4][256][4];
    const int lanes = C >> 2, tpb = 256 / lanes;
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes;
    const int th = H >> 1, tw = W >> 1;
    const size_t gstride = (size_t)T * C;
    const int n = cl * 4;
    const bool bnr = p.bnr_red1 != nullptr, bnr2 = bnr && p.bnr_red2 != nullptr;
    floatx4 bias = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = bias, mu1 = bias, is1 = bias, mu2 = bias, is2 = bias, msc = bias, msh = bias;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (p.bias) bias[c] = p.bias[n + c];
        if (p.scale) { sc[c] = p.scale[n + c]; sh[c] = p.shift[n + c]; }
        if (bnr) { mu1[c] = p.bnr_mean1[n + c]; is1[c] = p.bnr_invstd1[n + c]; }
        if (bnr && p.bnr_mscale) { msc[c] = p.bnr_mscale[n + c]; msh[c] = p.bnr_mshift[n + c]; }
        if (bnr2) { mu2[c] = p.bnr_mean2[n + c]; is2[c] = p.bnr_invstd2[n + c]; }
    }
    floatx4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        const float* src = Mt + (size_t)t * C + n;
        floatx4 m[4][4];
#pragma unroll
        for (int g = 0; g < 16; ++g) m[g >> 2][g & 3] = *reinterpret_cast<const floatx4*>(src + g * gstride);
        floatx4 a[2][4];                            // A^T m
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[0][j] = m[0][j] + m[1][j] + m[2][j];
            a[1][j] = m[1][j] - m[2][j] - m[3][j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                floatx4 v = j == 0 ? a[i][0] + a[i][1] + a[i][2] : a[i][1] - a[i][2] - a[i][3];
                v += bias;
                if (p.stats) { s0 += v; s1 += v * v; }
                v = v * sc + sh;
                const size_t o = ((size_t)(b * H + 2 * ty + i) * W + 2 * tx + j) * C + n;
                if (p.addend) v += *reinterpret_cast<const floatx4*>(p.addend + o);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                *reinterpret_cast<floatx4*>(p.y + o) = v;
                if (bnr) {
                    floatx4 dz;
                    const floatx4 y1 = *reinterpret_cast<const floatx4*>(p.bnr_y1 + o);
                    if (p.bnr_mask) {
                        const floatx4 mk = *reinterpret_cast<const floatx4*>(p.bnr_mask + o);
#pragma unroll
                        for (int c = 0; c < 4; ++c) dz[c] = mk[c] > 0.f ? v[c] : 0.f;
                    } else if (p.bnr_mask16) {
                        const ushort4 mk = *reinterpret_cast<const ushort4*>(p.bnr_mask16 + o);
                        dz[0] = (short)mk.x > 0 ? v[0] : 0.f; dz[1] = (short)mk.y > 0 ? v[1] : 0.f;
                        dz[2] = (short)mk.z > 0 ? v[2] : 0.f; dz[3] = (short)mk.w > 0 ? v[3] : 0.f;
                    } else {                         // the activation was never stored: its sign from the pre-BN output (common.h InBn)
#pragma unroll
                        for (int c = 0; c < 4; ++c) dz[c] = __builtin_fmaf(y1[c], msc[c], msh[c]) > 0.f ? v[c] : 0.f;
                    }
                    s0 += dz;
                    s1 += dz * ((y1 - mu1) * is1);
                    if (bnr2) {
                        const floatx4 y2 = *reinterpret_cast<const floatx4*>(p.bnr_y2 + o);
                        s2 += dz;
                        s3 += dz * ((y2 - mu2) * is2);
                    }
                }
            }
    }
    if (!p.stats && !bnr) return;                   // block-uniform
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        red[0][threadIdx.x][c] = s0[c]; red[1][threadIdx.x][c] = s1[c];
        red[2][threadIdx.x][c] = s2[c]; red[3][threadIdx.x][c] = s3[c];
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {  // channel ch lives in lane ch / 4, component ch % 4 of every tile lane group
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int k = 0; k < tpb; ++k) {
            const int src = k * lanes + (ch >> 2);
            a0 += (double)red[0][src][ch & 3]; a1 += (double)red[1][src][ch & 3];
            a2 += (double)red[2][src][ch & 3]; a3 += (double)red[3][src][ch & 3];
        }
        double* dst = p.stats ? p.stats : p.bnr_red1;
        unsafeAtomicAdd(dst + ch, a0);
        unsafeAtomicAdd(dst + C + ch, a1);
        if (bnr2) {
            unsafeAtomicAdd(p.bnr_red2 + ch, a2);
            unsafeAtomicAdd(p.bnr_red2 + C + ch, a3);
        }
    }
}

// ---- F(4x4,3x3) forward transforms (no-grad forwards) --------------------------------------------------------------------
// V4[g][t][c] = (B^T d B)[g / 6][g % 6] over 6x6 patches at stride 4; one thread = one tile x 4 channels
template <bool XBN>
__global__ void __launch_bounds__(256) wino4f_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C, int T,
                                                           const InBn in) {
    const int lanes = C >> 2, tpb = 256 / lanes;
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes;
    const int th = H >> 2, tw = W >> 2;
    const size_t gstride = (size_t)T * C;
    floatx4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XBN) inbn_coeff4(in, cl * 4, bsc, bsh);
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        const float* img = x + (size_t)b * H * W * C + cl * 4;
        floatx4 v[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {                // B^T d, column by column
            floatx4 a[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) a[i] = ld_tap<XBN>(img, y0 + i, x0 + j, H, W, C, bsc, bsh);
            w4f_bt(a);
            // (one column at a time: with all 36 unconditional loads hoisted the kernel needed 256 + 34 registers = one wave per SIMD)
            if constexpr (XBN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i][j] = a[i];
        }
        float* dst = V + (size_t)t * C + cl * 4;
#pragma unroll
        for (int i = 0; i < 6; ++i) {                // (B^T d) B, row by row
            w4f_bt(v[i]);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<floatx4*>(dst + (i * 6 + j) * gstride) = v[i][j];
        }
    }
    if constexpr (XBN) inbn_commit(in);
}

// y = A^T m A (4x4 outputs per tile) + the epilogue of igemm_epilogue.h (bias, BatchNorm batch statistics, folded-BN affine, residual,
// ReLU; for the dgrads -- backward.hip runs them in this form too -- the fused BatchNorm-backward sums of the layer(s) that consume the
// gradient next).  One thread = one tile x 4 channels: 2 / 1 channels per thread (2-4 x the waves of a B = 32 launch) measured SLOWER
// here and 4-6 % faster in the input transform, nothing on the step (profiles/r05_wino4f_channels_per_thread.txt).
__global__ void __launch_bounds__(256) wino4f_output_kernel(const float* __restrict__ Mt, const EpiArgs p, int B, int H, int W, int C, int T, int lanes) {
    __shared__ float red[4][256][4];
    // lanes = channel quads of this block's channel slice (blockIdx.y): see the launcher
    const int tpb = 256 / lanes;
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes;
    const int th = H >> 2, tw = W >> 2;
    const size_t gstride = (size_t)T * C;
    const int n = (blockIdx.y * lanes + cl) * 4;
    const bool bnr = p.bnr_red1 != nullptr, bnr2 = bnr && p.bnr_red2 != nullptr;
    floatx4 bias = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = bias, mu1 = bias, is1 = bias, mu2 = bias, is2 = bias, msc = bias, msh = bias;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (p.bias) bias[c] = p.bias[n + c];
        if (p.scale) { sc[c] = p.scale[n + c]; sh[c] = p.shift[n + c]; }
        if (bnr) { mu1[c] = p.bnr_mean1[n + c]; is1[c] = p.bnr_invstd1[n + c]; }
        if (bnr && p.bnr_mscale) { msc[c] = p.bnr_mscale[n + c]; msh[c] = p.bnr_mshift[n + c]; }
        if (bnr2) { mu2[c] = p.bnr_mean2[n + c]; is2[c] = p.bnr_invstd2[n + c]; }
    }
    floatx4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        const float* src = Mt + (size_t)t * C + n;
        floatx4 a[4][6];                             // A^T m, column by column of m
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            floatx4 m[6], y[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const floatx4*>(src + (i * 6 + j) * gstride);
            w4f_at(m, y);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][j] = y[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 y[4];
            w4f_at(a[i], y);                         // (A^T m) A, row by row
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                floatx4 v = y[j] + bias;
                if (p.stats) { s0 += v; s1 += v * v; }
                v = v * sc + sh;
                const size_t o = ((size_t)(b * H + 4 * ty + i) * W + 4 * tx + j) * C + n;
                if (p.addend) v += *reinterpret_cast<const floatx4*>(p.addend + o);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                *reinterpret_cast<floatx4*>(p.y + o) = v;
                if (bnr) {
                    floatx4 dz;
                    const floatx4 y1 = *reinterpret_cast<const floatx4*>(p.bnr_y1 + o);
                    if (p.bnr_mask) {
                        const floatx4 mk = *reinterpret_cast<const floatx4*>(p.bnr_mask + o);
#pragma unroll
                        for (int c = 0; c < 4; ++c) dz[c] = mk[c] > 0.f ? v[c] : 0.f;
                    } else if (p.bnr_mask16) {
                        const ushort4 mk = *reinterpret_cast<const ushort4*>(p.bnr_mask16 + o);
                        dz[0] = (short)mk.x > 0 ? v[0] : 0.f; dz[1] = (short)mk.y > 0 ? v[1] : 0.f;
                        dz[2] = (short)mk.z > 0 ? v[2] : 0.f; dz[3] = (short)mk.w > 0 ? v[3] : 0.f;
                    } else {                         // the activation was never stored: its sign from the pre-BN output (common.h InBn)
#pragma unroll
                        for (int c = 0; c < 4; ++c) dz[c] = __builtin_fmaf(y1[c], msc[c], msh[c]) > 0.f ? v[c] : 0.f;
                    }
                    s0 += dz;
                    s1 += dz * ((y1 - mu1) * is1);
                    if (bnr2) {
                        const floatx4 y2 = *reinterpret_cast<const floatx4*>(p.bnr_y2 + o);
                        s2 += dz;
                        s3 += dz * ((y2 - mu2) * is2);
                    }
                }
            }
        }
    }
    if (!p.stats && !bnr) return;                    // block-uniform
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        red[0][threadIdx.x][c] = s0[c]; red[1][threadIdx.x][c] = s1[c];
        red[2][threadIdx.x][c] = s2[c]; red[3][threadIdx.x][c] = s3[c];
    }
    __syncthreads();
    for (int lc = threadIdx.x; lc < 4 * lanes; lc += 256) {      // channel lc of the slice lives in lane lc / 4, component lc % 4 of every tile lane group
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int k = 0; k < tpb; ++k) {
            const int src = k * lanes + (lc >> 2);
            a0 += (double)red[0][src][lc & 3]; a1 += (double)red[1][src][lc & 3];
            a2 += (double)red[2][src][lc & 3]; a3 += (double)red[3][src][lc & 3];
        }
        const int ch = blockIdx.y * lanes * 4 + lc;
        double* dst = p.stats ? p.stats : p.bnr_red1;
        unsafeAtomicAdd(dst + ch, a0);
        unsafeAtomicAdd(dst + C + ch, a1);
        if (bnr2) {
            unsafeAtomicAdd(p.bnr_red2 + ch, a2);
            unsafeAtomicAdd(p.bnr_red2 + C + ch, a3);
        }
    }
}

// ---- weight gradient in the transform domain --------------------------------------------------------------------------
//   Y = A^T Mt A  =>  dMt = A dY A^T ;   Mt[g] = V[g] U[g]^T  =>  dU[g] = dMt[g]^T V[g] ;   U = G w G^T  =>  dw = G^T dU G
// dMt[g][t][c] = (A dY A^T)[g / 4][g % 4],  A = [[1,0],[1,1],[1,-1],[0,-1]]
__global__ void __launch_bounds__(256) wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ dM, int B, int H, int W, int C, int T) {
    const int lanes = C >> 2, tpb = 256 / lanes;
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes;
    const int th = H >> 1, tw = W >> 1;
    const size_t gstride = (size_t)T * C;
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        const float* src = dy + ((size_t)(b * H + 2 * ty) * W + 2 * tx) * C + cl * 4;
        const floatx4 y00 = *reinterpret_cast<const floatx4*>(src), y01 = *reinterpret_cast<const floatx4*>(src + C);
        const floatx4 y10 = *reinterpret_cast<const floatx4*>(src + (size_t)W * C), y11 = *reinterpret_cast<const floatx4*>(src + (size_t)W * C + C);
        floatx4 a[4][2];                            // A dY
        a[0][0] = y00; a[0][1] = y01;
        a[1][0] = y00 + y10; a[1][1] = y01 + y11;
        a[2][0] = y00 - y10; a[2][1] = y01 - y11;
        a[3][0] = -y10; a[3][1] = -y11;
        float* dst = dM + (size_t)t * C + cl * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 0) * gstride) = a[i][0];
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 1) * gstride) = a[i][0] + a[i][1];
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 2) * gstride) = a[i][0] - a[i][1];
            *reinterpret_cast<floatx4*>(dst + (i * 4 + 3) * gstride) = -a[i][1];
        }
    }
}

// Transposed-output forms for the weight gradient: out[g][c][t] (tile index contiguous), so that dU[g] = dMt[g]^T V[g] is a plain
// K-contiguous batched GEMM over K = T for the implicit-GEMM kernel (long K, no split / atomics).  A block covers 32 tiles x 32
// channels: reads are 128-B channel segments, the 16 transform elements go through an LDS tile [4 g][32 c][32 t] four at a time
// and leave as 128-B rows of 32 consecutive tiles.  DY = false: x -> (B^T d B)^T layout; DY = true: dy -> (A dY A^T).
template <bool DY, bool XBN = false>
__global__ void __launch_bounds__(256) wino_tr_kernel(const float* __restrict__ src, float* __restrict__ out, int B, int H, int W, int C, int T,
                                                      const InBn in) {
    static_assert(!(DY && XBN), "the BatchNorm-on-load form is for the activation operand");
    __shared__ float tbuf[4][32][33];
    const int tl = threadIdx.x >> 3, cq = threadIdx.x & 7;
    const int cblocks = C >> 5;
    const int t0 = (blockIdx.x / cblocks) * 32, c0 = (blockIdx.x % cblocks) * 32;
    const int t = t0 + tl, c = c0 + cq * 4;
    const int th = H >> 1, tw = W >> 1;
    floatx4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XBN) inbn_coeff4(in, c, bsc, bsh);
    floatx4 v[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) v[g] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (t < T) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        if constexpr (DY) {
            const float* p0 = src + ((size_t)(b * H + 2 * ty) * W + 2 * tx) * C + c;
            const floatx4 y00 = *reinterpret_cast<const floatx4*>(p0), y01 = *reinterpret_cast<const floatx4*>(p0 + C);
            const floatx4 y10 = *reinterpret_cast<const floatx4*>(p0 + (size_t)W * C), y11 = *reinterpret_cast<const floatx4*>(p0 + (size_t)W * C + C);
            floatx4 a[4][2];
            a[0][0] = y00; a[0][1] = y01;
            a[1][0] = y00 + y10; a[1][1] = y01 + y11;
            a[2][0] = y00 - y10; a[2][1] = y01 - y11;
            a[3][0] = -y10; a[3][1] = -y11;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i * 4 + 0] = a[i][0]; v[i * 4 + 1] = a[i][0] + a[i][1]; v[i * 4 + 2] = a[i][0] - a[i][1]; v[i * 4 + 3] = -a[i][1];
            }
        } else {
            const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
            const float* img = src + (size_t)b * H * W * C + c;
            floatx4 d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = ld_tap<XBN>(img, y0 + i, x0 + j, H, W, C, bsc, bsh);
            floatx4 e[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[0][j] = d[0][j] - d[2][j]; e[1][j] = d[1][j] + d[2][j]; e[2][j] = d[2][j] - d[1][j]; e[3][j] = d[1][j] - d[3][j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i * 4 + 0] = e[i][0] - e[i][2]; v[i * 4 + 1] = e[i][1] + e[i][2]; v[i * 4 + 2] = e[i][2] - e[i][1]; v[i * 4 + 3] = e[i][1] - e[i][3];
            }
        }
    }
    const int wr = threadIdx.x >> 5, wc = threadIdx.x & 31;        // write phase: row group / tile column
#pragma unroll
    for (int gp = 0; gp < 4; ++gp) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) tbuf[k][cq * 4 + e][tl] = v[gp * 4 + k][e];
        __syncthreads();
        if (t0 + wc < T) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wr + 8 * i, k = row >> 5, cc = row & 31;
                out[((size_t)(gp * 4 + k) * C + c0 + cc) * T + t0 + wc] = tbuf[k][cc][wc];
            }
        }
    }
}

// ---- F(4x4, 3x3) for the weight gradient only --------------------------------------------------------------------------
// 36 products per 4x4 outputs (2.25 per output instead of 4): 1.78x fewer matrix FLOPs again and 2.25x instead of 4x transform
// volume.  Its larger transform coefficients (up to 8, 1/24) cost ~10x the round-off of F(2x2,3x3) -- 6e-6 of the gradient's range in
// fp32 (simulated and measured, tests/test_gpu_ops.py) -- which is harmless HERE because a weight gradient is a leaf: nothing is
// propagated through it, whereas forward / dgrad errors would compound through eight layers (those stay F(2x2,3x3)).
//   dMt = A dY A^T (6x6 from 4x4),  V = B^T d B (6x6 patch, stride 4),  dU[g] = dMt[g] . V[g]^T over the tiles,  dw = G^T dU G
__device__ __forceinline__ void w4_bt(floatx4 (&a)[6]) {      // a <- B^T a
    const floatx4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    a[0] = 4.f * a0 - 5.f * a2 + a4;
    a[1] = -4.f * (a1 + a2) + a3 + a4;
    a[2] = 4.f * (a1 - a2) - a3 + a4;
    a[3] = 2.f * (a3 - a1) - a2 + a4;
    a[4] = 2.f * (a1 - a3) - a2 + a4;
    a[5] = 4.f * a1 - 5.f * a3 + a5;
}
__device__ __forceinline__ void w4_a(const floatx4 (&y)[4], floatx4 (&r)[6]) {   // r <- A y  (A = (A^T)^T, 6x4)
    const floatx4 e = y[0] + y[2], o = y[1] + y[3], e4 = y[0] + 4.f * y[2], o8 = 2.f * y[1] + 8.f * y[3];
    r[0] = y[0];
    r[1] = e + o;
    r[2] = e - o;
    r[3] = e4 + o8;
    r[4] = e4 - o8;
    r[5] = y[3];
}

template <bool DY, bool XBN = false>
__global__ void __launch_bounds__(256) wino4_tr_kernel(const float* __restrict__ src, float* __restrict__ out, int B, int H, int W, int C, int T,
                                                       const InBn in, int Kc, int S) {
    static_assert(!(DY && XBN), "the BatchNorm-on-load form is for the activation operand");
    __shared__ float tbuf[4][32][33];
    const int tl = threadIdx.x >> 3, cq = threadIdx.x & 7;
    const int cblocks = C >> 5;
    const int t0 = (blockIdx.x / cblocks) * 32, c0 = (blockIdx.x % cblocks) * 32;
    const int t = t0 + tl, c = c0 + cq * 4;
    const int th = H >> 2, tw = W >> 2;
    floatx4 bsc = {1.f, 1.f, 1.f, 1.f}, bsh = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XBN) inbn_coeff4(in, c, bsc, bsh);
    floatx4 v[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) v[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (t < T) {
        const int b = t / (th * tw), r = t - b * (th * tw);
        const int ty = r / tw, tx = r - ty * tw;
        if constexpr (DY) {
            floatx4 col[4][6];                       // A dY, column by column of dY
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                floatx4 y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = *reinterpret_cast<const floatx4*>(src + ((size_t)(b * H + 4 * ty + i) * W + 4 * tx + j) * C + c);
                w4_a(y, col[j]);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {            // (A dY) A^T, row by row
                const floatx4 y[4] = {col[0][i], col[1][i], col[2][i], col[3][i]};
                w4_a(y, v[i]);
            }
        } else {
            const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
            const float* img = src + (size_t)b * H * W * C + c;
#pragma unroll
            for (int j = 0; j < 6; ++j) {            // B^T d, column by column
                floatx4 a[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) a[i] = ld_tap<XBN>(img, y0 + i, x0 + j, H, W, C, bsc, bsh);
                w4_bt(a);
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i][j] = a[i];
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) w4_bt(v[i]); // (B^T d) B, row by row
        }
    }
    const int wr = threadIdx.x >> 5, wc = threadIdx.x & 31;
    // K-split of the contraction (S > 1): the tile range is cut into S chunks of Kc tiles (a multiple of this block's 32), chunk s of
    // transform element g is plane g * S + s, [C][Kc] -- the batched GEMM sees 36 S independent problems of depth Kc
    const int ks = t0 / Kc, kt = t0 - ks * Kc;
#pragma unroll
    for (int gp = 0; gp < 9; ++gp) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) tbuf[k][cq * 4 + e][tl] = v[(gp * 4 + k) / 6][(gp * 4 + k) % 6][e];
        __syncthreads();
        if (t0 + wc < T) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wr + 8 * i, k = row >> 5, cc = row & 31;
                out[(((size_t)(gp * 4 + k) * S + ks) * C + c0 + cc) * Kc + kt + wc] = tbuf[k][cc][wc];
            }
        }
    }
}

// dw[co][ky][kx][ci] = (G^T dU[.][co][ci] G)[ky][kx],  G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
__device__ __forceinline__ void w4_gt(const float (&u)[6], float (&s)[3]) {
    const float p = u[1] + u[2], m = u[2] - u[1], q = u[3] + u[4], n = u[3] - u[4];
    s[0] = 0.25f * u[0] - (1.f / 6.f) * p + (1.f / 24.f) * q;
    s[1] = (1.f / 6.f) * m + (1.f / 12.f) * n;
    s[2] = -(1.f / 6.f) * p + (1.f / 6.f) * q + u[5];
}
__global__ void __launch_bounds__(256) wino4_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int cout, int cin, int S) {
    const int total = cout * cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i / cin, ci = i - co * cin;
        float t[3][6];                              // G^T dU
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float u[6], s3[3];
#pragma unroll
            for (int r = 0; r < 6; ++r) {             // (the K-split's partial planes of an element, in chunk order)
                const float* pl = dU + (size_t)(r * 6 + c) * S * total + i;
                float acc = pl[0];
                for (int ks = 1; ks < S; ++ks) acc += pl[(size_t)ks * total];
                u[r] = acc;
            }
            w4_gt(u, s3);
            t[0][c] = s3[0]; t[1][c] = s3[1]; t[2][c] = s3[2];
        }
        float* o = dw + (size_t)co * 9 * cin + ci;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float s3[3];
            w4_gt(t[r], s3);
            o[(r * 3 + 0) * cin] = s3[0]; o[(r * 3 + 1) * cin] = s3[1]; o[(r * 3 + 2) * cin] = s3[2];
        }
    }
}

// dw[co][ky][kx][ci] = (G^T dU[.][co][ci] G)[ky][kx],  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
__global__ void __launch_bounds__(256) wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int cout, int cin) {
    const int total = cout * cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i / cin, ci = i - co * cin;
        float t[3][4];                              // G^T dU
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float u0 = dU[(size_t)(0 * 4 + c) * total + i], u1 = dU[(size_t)(1 * 4 + c) * total + i];
            const float u2 = dU[(size_t)(2 * 4 + c) * total + i], u3 = dU[(size_t)(3 * 4 + c) * total + i];
            t[0][c] = u0 + 0.5f * (u1 + u2);
            t[1][c] = 0.5f * (u1 - u2);
            t[2][c] = 0.5f * (u1 + u2) + u3;
        }
        float* o = dw + (size_t)co * 9 * cin + ci;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o[(r * 3 + 0) * cin] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            o[(r * 3 + 1) * cin] = 0.5f * (t[r][1] - t[r][2]);
            o[(r * 3 + 2) * cin] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        }
    }
}

}  // namespace

// Which convolutions of a plan run in Winograd form, and in which form, is a property of the PLAN: simq_plan_options
// (include/simq.h), read by layout.hip / forward.hip / backward.hip.  The functions below are pure geometry / cost rules.

// Geometry the kernels handle: 3x3 / stride 1 / pad 1 on an even-sized map, channel counts that fill the float4 lanes of the
// transform kernels (C / 4 divides 256) and the GEMM's tiles.
bool winograd_eligible(const ConvGeom& g) {
    auto lanes_ok = [](int c) { return c % 4 == 0 && c / 4 <= 256 && 256 % (c / 4) == 0; };
    return g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.Hin % 2 == 0 && g.Win % 2 == 0 && g.Hout == g.Hin && g.Wout == g.Win &&
           g.Cin % 16 == 0 && g.Cout % 64 == 0 && lanes_ok(g.Cin) && lanes_ok(g.Cout);
}

// Layers the transform pays for: enough multiply-adds per transformed element (tools/winograd_probe.py); the Cin * Cout threshold
// is simq_plan_options.winograd_min_cc (default 128 * 128).  Round 1 (F(2x2,3x3) only) drew the line at 128 * 256; with the F(4x4,3x3) forms of the
// no-grad forwards and dgrads the 128 -> 128 layers of layer2 pay as well: 3131 -> 3175 tr/s on configs[1] (forward + backward alone
// unchanged: the gain is in the two no-grad forwards); 64 -> 64 (layer1) does not (3155, forward + backward -0.7 %).
bool winograd_pays(int cin, int cout, long min_cc) { return (long)cin * cout >= min_cc; }

// grid of a grid-stride kernel: at most `cap` blocks, every block the same number of trips (a capped grid whose last trip is ragged
// runs as long as one more full trip)
static int balanced_grid(int blocks, int cap) {
    if (blocks <= cap) return blocks < 1 ? 1 : blocks;
    const int trips = (blocks + cap - 1) / cap;
    return (blocks + trips - 1) / trips;
}

int64_t winograd_scratch_floats(const ConvGeom& g) {
    const int64_t T = (int64_t)g.B * (g.Hin / 2) * (g.Win / 2);
    return 16 * T * (g.Cin + g.Cout);
}

int launch_wino_weight(const float* w_ohwi, float* U, int cout, int cin, hipStream_t stream) {
    const int total = cout * cin;
    int blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wino_weight_kernel, dim3(blocks), dim3(256), 0, stream, w_ohwi, U, cout, cin);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

int launch_wino_weight_all(const float* params, const float* wt, float* ubase, const WinoWeightTable& t, hipStream_t stream) {
    if (t.n == 0) return 0;
    hipLaunchKernelGGL(wino_weight_all_kernel, dim3(256, (unsigned)t.n), dim3(256), 0, stream, params, wt, ubase, t);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

// scratch: winograd_scratch_floats(g) floats (V | Mt), 16-byte aligned
int launch_conv_winograd(const float* x, const float* U, float* y, const ConvGeom& g, const ConvEpilogue& e, float* scratch,
                         hipStream_t stream, const InBn& in) {
    SIMQ_REQUIRE(winograd_eligible(g), "conv_winograd: geometry not supported (3x3 s1 p1, even map, Cin %% 16, Cout %% 64)");
    const int T = g.B * (g.Hin / 2) * (g.Win / 2);
    SIMQ_REQUIRE((double)T * 16 * (g.Cin > g.Cout ? g.Cin : g.Cout) * 4.0 < 68719476736.0, "conv_winograd: batch too large");
    float* V = scratch;
    float* Mt = scratch + (size_t)16 * T * g.Cin;
    const int tpb_in = 256 / (g.Cin / 4), tpb_out = 256 / (g.Cout / 4);
    int bin = (T + tpb_in - 1) / tpb_in;
    if (bin > 4096) bin = 4096;
    if (in.on()) hipLaunchKernelGGL(wino_input_kernel<true>, dim3(bin), dim3(256), 0, stream, x, V, g.B, g.Hin, g.Win, g.Cin, T, in);
    else hipLaunchKernelGGL(wino_input_kernel<false>, dim3(bin), dim3(256), 0, stream, x, V, g.B, g.Hin, g.Win, g.Cin, T, in);
    SIMQ_CHECK_LAUNCH();
    if (int rc = launch_gemm_batched(V, U, Mt, T, g.Cout, g.Cin, 16, stream, g.tune)) return rc;
    int bout = (T + tpb_out - 1) / tpb_out;
    // statistics: few blocks, one fp64 atomic per channel and block (the step is insensitive to this cap from 256 to 2048)
    bout = balanced_grid(bout, (e.stats || e.bnr_red1) ? 512 : 4096);
    const EpiArgs ea = make_epi(y, e);
    note_launch("winograd_f2");
    hipLaunchKernelGGL(wino_output_kernel, dim3(bout), dim3(256), 0, stream, Mt, ea, g.B, g.Hin, g.Win, g.Cout, T);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

// F(4x4,3x3) form for the no-grad forwards: U4 = G4 w G4^T [36][Cout][Cin] (weight cache); scratch as for the F(2x2,3x3) form
// (V4 | Mt4 = 9 * T * (Cin + Cout) floats fit the 16 * T * (Cin + Cout) region).  Eligible maps are multiples of 4.
bool winograd_f4_forward(const ConvGeom& g, int min_tiles) {      // (simq_plan_options.winograd_f4_forward / _f4_min_tiles)
    return winograd_eligible(g) && g.Hin % 4 == 0 && g.Win % 4 == 0 && g.B * (g.Hin / 4) * (g.Win / 4) >= min_tiles;
}

// How far F(4x4,3x3) reaches into the differentiated path (simq_plan_options.winograd_f4_grad):
//   2 (default)  the DGRADS too.  The gradient-parity study (tests/test_gpu_fcn.py::test_gradient_parity_distribution, 13 seeded
//                batches vs fp64) is unchanged to three digits by it -- median error 1.61e-3 either way (reference fp32: 2.04e-3):
//                the error of these gradients is made by the FORWARD's round-off (ReLU masks, x-hat of the train-mode BatchNorms),
//                a dgrad's own 3e-6 is not amplified.  Step 2899 -> 3051 tr/s.
//   1            the grad-mode forward as well: 3231 tr/s, but the median gradient error doubles to 3.1e-3 (1.5 x the reference's
//                own fp32) -- within the study's bar, not adopted: the forward's accuracy is the gradient's accuracy.
//   0            the differentiated path stays F(2x2,3x3) entirely.
// simq_plan_options.winograd_f4_fwd_grad_min_cc (default 512*512) adds the GRAD-MODE forward of the layers with Cin*Cout at or above it:
// layer4's three 512->512 convolutions -- their round-off passes through no further residual block and leaves the gradient study
// (19 batches) unchanged, +4.4 % on the step; from 256->512 down the study's tail grows (docs/history.md 4, tests/diag_f4_grad_layers.py).

int launch_conv_winograd4(const float* x, const float* U4, float* y, const ConvGeom& g, const ConvEpilogue& e, float* scratch,
                          hipStream_t stream, const InBn& in) {
    SIMQ_REQUIRE(winograd_eligible(g) && g.Hin % 4 == 0 && g.Win % 4 == 0, "conv_winograd4: geometry not supported");
    SIMQ_REQUIRE(!e.y_bf16, "conv_winograd4: fp32 outputs only");
    const int T4 = g.B * (g.Hin / 4) * (g.Win / 4);
    float* V = scratch;
    float* Mt = scratch + (size_t)36 * T4 * g.Cin;
    const int tpb_in = 256 / (g.Cin / 4);
    int bin = (T4 + tpb_in - 1) / tpb_in;
    if (bin > 4096) bin = 4096;
    if (in.on()) hipLaunchKernelGGL(wino4f_input_kernel<true>, dim3(bin), dim3(256), 0, stream, x, V, g.B, g.Hin, g.Win, g.Cin, T4, in);
    else hipLaunchKernelGGL(wino4f_input_kernel<false>, dim3(bin), dim3(256), 0, stream, x, V, g.B, g.Hin, g.Win, g.Cin, T4, in);
    SIMQ_CHECK_LAUNCH();
    if (int rc = launch_gemm_batched(V, U4, Mt, T4, g.Cout, g.Cin, 36, stream, g.tune)) return rc;
    // Round 6.  A launch that leaves BatchNorm sums (train-mode statistics; the dgrads' fused backward sums) ends with 2-4 fp64 atomics per
    // channel and block, and its blocks all finish together: 576 blocks x 1024 sums onto 64 cache lines serialise (~6 ns each) into tens of
    // microseconds -- the 512-channel launches took 32-95 us where the bare transform takes 19.  So such a launch cuts the channels into slices
    // of `lanes` quads (blockIdx.y) and gives a block MORE TILES of fewer channels: the same blocks and bytes, 1 / (C / 4 / lanes) of the
    // atomics per block (256-byte instead of 2-KB runs per tile and plane: still whole DRAM bursts).
    static const int slice_quads = SIMQ_TUNE_INT("SIMQ_W4F_OUT_SLICE_QUADS", 16);      // (ablation build: 0 = whole channel range per block, rounds 2-5)
    int lanes_out = g.Cout / 4;
    if ((e.stats || e.bnr_red1) && slice_quads > 0 && lanes_out > slice_quads && lanes_out % slice_quads == 0) lanes_out = slice_quads;
    const int tpb_o = 256 / lanes_out, slices = (g.Cout / 4) / lanes_out;
    int bout = (T4 + tpb_o - 1) / tpb_o;
    // F(4x4,3x3) has a quarter of the tiles: at 512 channels a block is two tiles and B = 32 gives 576 blocks -- capped at 512, sixty-four
    // blocks did two grid-stride trips while the rest did one (the launch took the time of two).  Up to 1024 blocks, equal trips each.
    bout = balanced_grid(bout, ((e.stats || e.bnr_red1) ? 1024 : 4096) / slices);
    const EpiArgs ea = make_epi(y, e);
    note_launch("winograd_f4");
    hipLaunchKernelGGL(wino4f_output_kernel, dim3(bout, slices), dim3(256), 0, stream, Mt, ea, g.B, g.Hin, g.Win, g.Cout, T4, lanes_out);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

// Weight gradient of an eligible convolution through the transform domain: dw (OHWI, overwritten) from x and dy.
// scratch: winograd_scratch_floats(g) + 36*Cout*Cin floats (V | dMt | dU; the F(4x4,3x3) form needs 9*T*(Cin+Cout) + 36*Cout*Cin).
// Cin % 128 == 0 and Cout % 128 == 0.
bool winograd_wgrad_eligible(const ConvGeom& g) { return winograd_eligible(g) && g.Cin % 128 == 0 && g.Cout % 128 == 0; }

bool winograd_wgrad_f4(const ConvGeom& g) {   // the F(4x4,3x3) form applies to this geometry (simq_plan_options.winograd_wgrad_f4 allows it)
    return g.Hin % 4 == 0 && g.Win % 4 == 0 && (g.B * (g.Hin / 4) * (g.Win / 4)) % 16 == 0;
}

// where the transform domain beats the direct wgrad kernel (tools/winograd_probe.py): from 256 x 256 channels with F(2x2,3x3),
// from 128 x 256 with F(4x4,3x3) (0.081 vs 0.118 ms; F(2x2,3x3) loses there)
bool winograd_wgrad_pays(const ConvGeom& g, bool allow_f4) {
    return (long)g.Cin * g.Cout >= ((allow_f4 && winograd_wgrad_f4(g)) ? 128L * 256 : 256L * 256);
}

int launch_conv_wgrad_winograd(const float* x, const float* dy, float* dw, const ConvGeom& g, float* scratch, hipStream_t stream, bool allow_f4,
                               const InBn& in) {
    SIMQ_REQUIRE(winograd_wgrad_eligible(g), "conv_wgrad_winograd: geometry not supported");
    const int T = g.B * (g.Hin / 2) * (g.Win / 2);
    float* Vt = scratch;                                   // [16][Cin][T]
    float* dMt = scratch + (size_t)16 * T * g.Cin;         // [16][Cout][T]
    float* dU = dMt + (size_t)16 * T * g.Cout;             // [16][Cout][Cin]
    static const int direct_form = SIMQ_TUNE_INT("SIMQ_WINOGRAD_WGRAD_SPLITK", 0);     // (kernel-form ablation)
    const int T4 = g.B * (g.Hin / 4) * (g.Win / 4);
    if (!direct_form && allow_f4 && winograd_wgrad_f4(g)) {   // F(4x4,3x3): 36 GEMMs over T4 = T / 4 tiles
        // K-split (round 4): 36 GEMMs of (Cout / 64) x (Cin / 64) tiles are 576 blocks at 256 x 256 channels -- 2.25 per CU where the
        // 64 x 64 tile wants 6-8 -- over a contraction of T4 = 1152+ tiles.  S chunks of the tile range become 36 S planes (the transposing
        // transforms write them, wino4_dw_kernel adds them in chunk order: deterministic); the partial planes use the part of the
        // documented scratch the F(4x4,3x3) operands leave free.
        int S = 1;
        {
            const int ksplit = (g.tune.wgrad_ksplit == 1 || g.tune.wgrad_ksplit == 2 || g.tune.wgrad_ksplit == 4) ? g.tune.wgrad_ksplit : 0;   // 0: by shape
            const long blocks = 36L * (g.Cout / 64) * (g.Cin / 64);
            const long room = (long)winograd_scratch_floats(g) - 36L * T4 * (g.Cin + g.Cout);      // floats behind the operands, beside dU4's own 36 planes
            for (int cand = 4; cand >= 2; cand >>= 1) {
                if (ksplit > 0 && cand != ksplit) continue;
                const bool fits = T4 % (cand * 32) == 0 && 36L * (cand - 1) * g.Cout * g.Cin <= room;
                // by shape (tools/wgrad_ksplit_check.py, whole launch sequence, S = 1 | 2 | 4): B = 32 256 -> 256 108 | 104 | 119 us,
                // 128 -> 256 87 | 83 | 94; B = 128 256 -> 256 370 | 332 | 338, 128 -> 256 268 | 231 | 229, 256 -> 512 555 | 526 | 548;
                // 512 -> 512 (2304 blocks) loses at every batch size -- shorter chains only pay while the CUs are short of blocks
                const int kc = T4 / cand;
                const bool pays = cand == 4 ? (blocks <= 288 && kc >= 1152) : ((blocks <= 576 && kc >= 512) || (blocks <= 1152 && kc >= 1152));
                if (fits && (ksplit > 0 || pays)) { S = cand; break; }
            }
            if (ksplit == 1) S = 1;
        }
        const int Kc = T4 / S;
        float* Vt4 = scratch;                                   // [36][S][Cin][Kc]
        float* dMt4 = scratch + (size_t)36 * T4 * g.Cin;        // [36][S][Cout][Kc]
        float* dU4 = dMt4 + (size_t)36 * T4 * g.Cout;           // [36][S][Cout][Cin]
        const int tb = (T4 + 31) / 32;
        if (in.on()) hipLaunchKernelGGL((wino4_tr_kernel<false, true>), dim3((unsigned)(tb * (g.Cin / 32))), dim3(256), 0, stream, x, Vt4, g.B, g.Hin, g.Win, g.Cin, T4, in, Kc, S);
        else hipLaunchKernelGGL((wino4_tr_kernel<false>), dim3((unsigned)(tb * (g.Cin / 32))), dim3(256), 0, stream, x, Vt4, g.B, g.Hin, g.Win, g.Cin, T4, in, Kc, S);
        hipLaunchKernelGGL((wino4_tr_kernel<true>), dim3((unsigned)(tb * (g.Cout / 32))), dim3(256), 0, stream, dy, dMt4, g.B, g.Hin, g.Win, g.Cout, T4, InBn(), Kc, S);
        SIMQ_CHECK_LAUNCH();
        if (int rc = launch_gemm_batched(dMt4, Vt4, dU4, g.Cout, g.Cin, Kc, 36 * S, stream, g.tune)) return rc;
        int blocks4 = (g.Cout * g.Cin + 255) / 256;
        if (blocks4 > 2048) blocks4 = 2048;
        note_launch("winograd_f4_wgrad");
        hipLaunchKernelGGL(wino4_dw_kernel, dim3(blocks4), dim3(256), 0, stream, dU4, dw, g.Cout, g.Cin, S);
        SIMQ_CHECK_LAUNCH();
        return 0;
    }
    if (direct_form || T % 16 != 0) {   // [g][t][c] operands, pixel-split batched wgrad_kernel with fp32 atomics: tile counts that are
                                        // not a multiple of the GEMM's K-step (never the 24x24 maps of the network), and A-B runs
        const int tpb_in = 256 / (g.Cin / 4), tpb_out = 256 / (g.Cout / 4);
        int bin = (T + tpb_in - 1) / tpb_in, bout = (T + tpb_out - 1) / tpb_out;
        if (bin > 4096) bin = 4096;
        if (bout > 4096) bout = 4096;
        if (in.on()) hipLaunchKernelGGL(wino_input_kernel<true>, dim3(bin), dim3(256), 0, stream, x, Vt, g.B, g.Hin, g.Win, g.Cin, T, in);
        else hipLaunchKernelGGL(wino_input_kernel<false>, dim3(bin), dim3(256), 0, stream, x, Vt, g.B, g.Hin, g.Win, g.Cin, T, in);
        hipLaunchKernelGGL(wino_dy_kernel, dim3(bout), dim3(256), 0, stream, dy, dMt, g.B, g.Hin, g.Win, g.Cout, T);
        SIMQ_CHECK_LAUNCH();
        SIMQ_CHECK_HIP(hipMemsetAsync(dU, 0, sizeof(float) * 16 * (size_t)g.Cout * g.Cin, stream));
        if (int rc = launch_wgrad_batched(Vt, dMt, dU, T, g.Cout, g.Cin, 16, stream, g.tune)) return rc;
    } else {
        // tile index contiguous: dU[g] = dMt[g] (Cout x T) * Vt[g]^T (T x Cin) is a K-contiguous GEMM with K = T
        const int tb = (T + 31) / 32;
        if (in.on()) hipLaunchKernelGGL((wino_tr_kernel<false, true>), dim3((unsigned)(tb * (g.Cin / 32))), dim3(256), 0, stream, x, Vt, g.B, g.Hin, g.Win, g.Cin, T, in);
        else hipLaunchKernelGGL((wino_tr_kernel<false>), dim3((unsigned)(tb * (g.Cin / 32))), dim3(256), 0, stream, x, Vt, g.B, g.Hin, g.Win, g.Cin, T, in);
        hipLaunchKernelGGL((wino_tr_kernel<true>), dim3((unsigned)(tb * (g.Cout / 32))), dim3(256), 0, stream, dy, dMt, g.B, g.Hin, g.Win, g.Cout, T, InBn());
        SIMQ_CHECK_LAUNCH();
        if (int rc = launch_gemm_batched(dMt, Vt, dU, g.Cout, g.Cin, T, 16, stream, g.tune)) return rc;
    }
    int blocks = (g.Cout * g.Cin + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    note_launch("winograd_f2_wgrad");
    hipLaunchKernelGGL(wino_dw_kernel, dim3(blocks), dim3(256), 0, stream, dU, dw, g.Cout, g.Cin);
    SIMQ_CHECK_LAUNCH();
    return 0;
}

}  // namespace simq
