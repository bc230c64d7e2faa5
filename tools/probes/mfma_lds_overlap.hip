// Probe: do ds_read_b128 streams overlap with bf16 MFMA issue on gfx950?  Per iteration each wave issues NM independent
// 16x16x32 MFMAs and ND conflict-free ds_read_b128 (results consumed only at the end of the iteration).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NM, int ND>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    floatx4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = floatx4{0, 0, 0, 0};
    const char* base = smem + (wave & 3) * 8192 + lane * 16;     // lane-linear 1 KB per read: conflict-free
    bf16x8 f[ND > 0 ? ND : 1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < ND; ++d) f[d] = *reinterpret_cast<const bf16x8*>(base + (d % 8) * 1024);
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 12], 0, 0, 0);
        if (ND > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int d = 0; d < ND; ++d) asm volatile("" :: "v"(f[d]));
        }
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int ND>
void run(float* out, int waves) {
    const int iters = 3000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, ND>), dim3(256), dim3(waves * 64), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, ND>), dim3(256), dim3(waves * 64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves/CU=%d  MFMA=%2d  ds_read_b128=%2d per iteration : %7.1f ns/iter  (MFMA alone would be %.0f ns at 16 clk@2.2GHz x %d waves/SIMD)\n",
           waves, NM, ND, ms * 1e6 / iters, NM * 16 / 2.2 * (waves / 4), waves / 4);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int w : {4, 8}) {
        run<36, 0>(out, w); run<0, 13>(out, w); run<36, 13>(out, w); run<36, 26>(out, w); run<0, 26>(out, w); run<18, 11>(out, w); run<0, 11>(out, w);
    }
    return 0;
}
