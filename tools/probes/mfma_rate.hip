// Probe: issue-rate of the gfx950 bf16 MFMA shapes with independent accumulators (1, 2, 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    if constexpr (SHAPE == 16) {
        floatx4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        floatx16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE, int NACC>
void run(float* out, int waves_per_cu) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(waves_per_cu * 64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(waves_per_cu * 64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops_per = SHAPE == 16 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
    const double total = flops_per * NACC * iters * waves_per_cu * 256;
    const double per_simd_ns = ms * 1e6 / ((double)NACC * iters * (waves_per_cu / 4.0));
    printf("mfma %s  nacc=%2d waves/CU=%2d : %7.1f TF/s   %.2f ns per MFMA per SIMD (%.1f clk @2.4GHz)\n", SHAPE == 16 ? "16x16x32" : "32x32x16",
           NACC, waves_per_cu, total / (ms * 1e9), per_simd_ns, per_simd_ns * 2.4);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w : {4, 8, 16}) {
        run<16, 4>(out, w); run<16, 12>(out, w); run<16, 36>(out, w);
        run<32, 2>(out, w); run<32, 4>(out, w); run<32, 8>(out, w);
    }
    return 0;
}
