// Probe (GPU): the F(4x4,3x3) OUTPUT transform of one convolution fused with the INPUT transform of the next one -- the smallest honest form of
// "fused Winograd" for the one place where nothing grid-wide stands between the two: an eval-mode forward (target net; BatchNorm folded into a
// per-channel scale / shift), conv1 -> conv2 inside a BasicBlock (resnet.py:34-40), whose activation in between has no other reader.
//
//   two launches (what the plan runs):   Mt[36][T][C] -> y = relu((A^T m A) * scale + shift)  [B][24][24][C]      wino4f_output_kernel
//                                        y -> V[36][T][C] = B^T d B over 6x6 patches at stride 4, pad 1            wino4f_input_kernel
//   fused (this probe):                  one block per (image, 32-channel slice): the 24 x 24 x 32 activation lives in LDS (73.7 KB) between
//                                        the two stages and never reaches HBM
// Same arithmetic in the same order per element: the fused V must equal the two-launch V bit for bit (checked).  Prints us per launch and
// the algorithmic bytes of both forms for B = 32 / 29 and C = 128 / 256 / 512.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/fused_transform_probe.hip -o /tmp/ftp && /tmp/ftp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// interpolation points {0, 1, -1, 1/2, -2, inf} (conv_winograd.hip):  A^T = [[1,1,1,1,1,0],[0,1,-1,1/2,-2,0],[0,1,1,1/4,4,0],[0,1,-1,1/8,-8,1]]
__device__ __forceinline__ void at4(const floatx4 (&m)[6], floatx4 (&y)[4]) {
    const floatx4 s = m[1] + m[2], d = m[1] - m[2];
    y[0] = m[0] + s + m[3] + m[4];
    y[1] = d + 0.5f * m[3] - 2.f * m[4];
    y[2] = s + 0.25f * m[3] + 4.f * m[4];
    y[3] = d + 0.125f * m[3] - 8.f * m[4] + m[5];
}
// B^T = [[1,-3/2,-2,3/2,1,0],[0,-1,1/2,5/2,1,0],[0,1,-5/2,1/2,1,0],[0,-2,-1,2,1,0],[0,1/2,-1,-1/2,1,0],[0,1,-3/2,-2,3/2,1]]
__device__ __forceinline__ void bt4(floatx4 (&a)[6]) {
    const floatx4 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4], a5 = a[5];
    a[0] = a0 + 1.5f * (a3 - a1) - 2.f * a2 + a4;
    a[1] = 0.5f * a2 - a1 + 2.5f * a3 + a4;
    a[2] = a1 - 2.5f * a2 + 0.5f * a3 + a4;
    a[3] = 2.f * (a3 - a1) - a2 + a4;
    a[4] = 0.5f * (a1 - a3) - a2 + a4;
    a[5] = a1 - 1.5f * a2 - 2.f * a3 + 1.5f * a4 + a5;
}
__device__ __forceinline__ floatx4 relu_affine(floatx4 v, floatx4 sc, floatx4 sh) {
    v = v * sc + sh;
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    return v;
}

// ---- the two launches: one thread = one tile x 4 channels (the plan's decomposition) ----
__global__ void __launch_bounds__(256) out_kernel(const float* __restrict__ Mt, const float* __restrict__ scale, const float* __restrict__ shift,
                                                  float* __restrict__ y, int T, int C) {
    const int lanes = C >> 2, tpb = 256 / lanes;
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes, n = cl * 4;
    const size_t gs = (size_t)T * C;
    const floatx4 sc = *reinterpret_cast<const floatx4*>(scale + n), sh = *reinterpret_cast<const floatx4*>(shift + n);
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / 36, r = t - b * 36, ty = r / 6, tx = r - ty * 6;
        const float* src = Mt + (size_t)t * C + n;
        floatx4 a[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            floatx4 m[6], o[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const floatx4*>(src + (i * 6 + j) * gs);
            at4(m, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][j] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 o[4];
            at4(a[i], o);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<floatx4*>(y + ((size_t)(b * 24 + 4 * ty + i) * 24 + 4 * tx + j) * C + n) = relu_affine(o[j], sc, sh);
        }
    }
}
__global__ void __launch_bounds__(256) in_kernel(const float* __restrict__ x, float* __restrict__ V, int T, int C) {
    const int lanes = C >> 2, tpb = 256 / lanes;
    const int cl = threadIdx.x % lanes, tl = threadIdx.x / lanes, n = cl * 4;
    const size_t gs = (size_t)T * C;
    for (int t = blockIdx.x * tpb + tl; t < T; t += gridDim.x * tpb) {
        const int b = t / 36, r = t - b * 36, ty = r / 6, tx = r - ty * 6;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        floatx4 v[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            floatx4 a[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int yy = y0 + i, xx = x0 + j;
                const bool ok = (unsigned)yy < 24u && (unsigned)xx < 24u;
                a[i] = ok ? *reinterpret_cast<const floatx4*>(x + ((size_t)(b * 24 + yy) * 24 + xx) * C + n) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
            bt4(a);
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i][j] = a[i];
        }
        float* dst = V + (size_t)t * C + n;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            bt4(v[i]);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<floatx4*>(dst + (i * 6 + j) * gs) = v[i][j];
        }
    }
}

// ---- fused: one block = one image x SLICE channels; the activation stays in LDS ----
template <int SLICE>
__global__ void __launch_bounds__(256) fused_kernel(const float* __restrict__ Mt, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    float* __restrict__ V, int T, int C) {
    extern __shared__ float act[];                       // [24][24][SLICE]
    constexpr int Q = SLICE / 4;                          // channel quads of the slice
    const int slices = C / SLICE;
    const int b = blockIdx.x / slices, c0 = (blockIdx.x % slices) * SLICE;
    const size_t gs = (size_t)T * C;
    for (int w = threadIdx.x; w < 36 * Q; w += 256) {    // stage 1: output transform of the image's 36 tiles into LDS
        const int q = w % Q, r = w / Q, ty = r / 6, tx = r - ty * 6, n = c0 + q * 4;
        const floatx4 sc = *reinterpret_cast<const floatx4*>(scale + n), sh = *reinterpret_cast<const floatx4*>(shift + n);
        const float* src = Mt + (size_t)(b * 36 + r) * C + n;
        floatx4 a[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            floatx4 m[6], o[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const floatx4*>(src + (i * 6 + j) * gs);
            at4(m, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][j] = o[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 o[4];
            at4(a[i], o);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<floatx4*>(act + ((4 * ty + i) * 24 + 4 * tx + j) * SLICE + q * 4) = relu_affine(o[j], sc, sh);
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < 36 * Q; w += 256) {    // stage 2: input transform of the next convolution out of LDS
        const int q = w % Q, r = w / Q, ty = r / 6, tx = r - ty * 6, n = c0 + q * 4;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        floatx4 v[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            floatx4 a[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int yy = y0 + i, xx = x0 + j;
                const bool ok = (unsigned)yy < 24u && (unsigned)xx < 24u;
                a[i] = ok ? *reinterpret_cast<const floatx4*>(act + (yy * 24 + xx) * SLICE + q * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
            bt4(a);
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i][j] = a[i];
        }
        float* dst = V + (size_t)(b * 36 + r) * C + n;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            bt4(v[i]);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<floatx4*>(dst + (i * 6 + j) * gs) = v[i][j];
        }
    }
}

template <typename F>
static float time_us(F launch, int iters = 50) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main() {
    // (clocks: a second of work first)
    for (int B : {32, 29})
        for (int C : {128, 256, 512}) {
            const int T = B * 36;
            const size_t plane = (size_t)36 * T * C, act = (size_t)B * 576 * C;
            float *Mt, *y, *V0, *V1, *sc, *sh;
            CK(hipMalloc(&Mt, plane * 4)); CK(hipMalloc(&V0, plane * 4)); CK(hipMalloc(&V1, plane * 4)); CK(hipMalloc(&y, act * 4));
            CK(hipMalloc(&sc, C * 4)); CK(hipMalloc(&sh, C * 4));
            std::vector<float> h(plane), hs(C), hb(C);
            unsigned s = 12345u + C + B;
            for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
            for (int c = 0; c < C; ++c) { hs[c] = 0.5f + 0.001f * c; hb[c] = 0.01f * (c % 7) - 0.02f; }
            CK(hipMemcpy(Mt, h.data(), plane * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(sc, hs.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, hb.data(), C * 4, hipMemcpyHostToDevice));
            const int tpb = 256 / (C / 4);
            const int blocks = (T + tpb - 1) / tpb;
            auto two = [&]() {
                hipLaunchKernelGGL(out_kernel, dim3(blocks), dim3(256), 0, 0, Mt, sc, sh, y, T, C);
                hipLaunchKernelGGL(in_kernel, dim3(blocks), dim3(256), 0, 0, y, V0, T, C);
            };
            auto f32s = [&]() { hipLaunchKernelGGL(fused_kernel<32>, dim3(B * (C / 32)), dim3(256), 24 * 24 * 32 * 4, 0, Mt, sc, sh, V1, T, C); };
            auto f64s = [&]() { hipLaunchKernelGGL(fused_kernel<64>, dim3(B * (C / 64)), dim3(256), 24 * 24 * 64 * 4, 0, Mt, sc, sh, V1, T, C); };
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 24 * 64 * 4));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 24 * 32 * 4));
            if (B == 32 && C == 128) for (int i = 0; i < 400; ++i) two();          // warm the clocks once
            const float t_out = time_us([&]() { hipLaunchKernelGGL(out_kernel, dim3(blocks), dim3(256), 0, 0, Mt, sc, sh, y, T, C); });
            const float t_in = time_us([&]() { hipLaunchKernelGGL(in_kernel, dim3(blocks), dim3(256), 0, 0, y, V0, T, C); });
            const float t_two = time_us(two);
            const float t_f32 = time_us(f32s);
            std::vector<float> a(plane), b2(plane);
            CK(hipMemcpy(a.data(), V0, plane * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), V1, plane * 4, hipMemcpyDeviceToHost));
            const bool same32 = memcmp(a.data(), b2.data(), plane * 4) == 0;
            const float t_f64 = time_us(f64s);
            CK(hipMemcpy(b2.data(), V1, plane * 4, hipMemcpyDeviceToHost));
            const bool same64 = memcmp(a.data(), b2.data(), plane * 4) == 0;
            const double mb_two = (2.0 * plane + 2.0 * act) * 4 / 1e6, mb_fused = 2.0 * plane * 4 / 1e6;
            printf("B=%2d C=%3d   two launches: out %6.1f + in %6.1f us, back to back %6.1f us (%6.1f MB, %.2f TB/s)   fused 32-ch slices %6.1f us%s   64-ch slices %6.1f us%s   "
                   "(%6.1f MB, %.2f TB/s at the better)   fused / two = %.2f\n", B, C, t_out, t_in, t_two, mb_two, mb_two / t_two,
                   t_f32, same32 ? "" : " [DIFFERS]", t_f64, same64 ? "" : " [DIFFERS]", mb_fused, mb_fused / (t_f32 < t_f64 ? t_f32 : t_f64), (t_f32 < t_f64 ? t_f32 : t_f64) / t_two);
            CK(hipFree(Mt)); CK(hipFree(V0)); CK(hipFree(V1)); CK(hipFree(y)); CK(hipFree(sc)); CK(hipFree(sh));
        }
    return 0;
}
