// Probe: global -> LDS fill rate per CU on gfx950 for (a) LDS-DMA buffer loads (16 B / lane) and (b) buffer loads to
// VGPRs + ds_write_b128, as a function of waves per CU and instructions in flight per wave.  Sources are L2-resident.
// T(D) = latency + D * per-instruction cost; one block per CU (LDS-limited), 256 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;

template <int D, int MODE>
__global__ void __launch_bounds__(1024) fill(const uint32_t* src, unsigned bytes, int iters, uint32_t* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[96 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, bytes, 0x00020000);
    // each wave owns D KB of LDS (wave * D KB); full 128-B lines: 8 lanes per line, lines 1 KB apart (like NHWC pixels)
    unsigned voff = (unsigned)((lane >> 3) * 1024 + (lane & 7) * 16 + wave * 8192 + blockIdx.x * 65536);
    char* base = smem + (wave * D % 96) * 1024;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(base + (d % 8) * 1024), 16, (voff + d * 128) & (bytes - 1), 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 v[D];
#pragma unroll
            for (int d = 0; d < D; ++d)
                v[d] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (voff + d * 128) & (bytes - 1), 0, 0));
#pragma unroll
            for (int d = 0; d < D; ++d) *reinterpret_cast<uint4*>(base + (d % 8) * 1024 + lane * 16) = v[d];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        voff += 16384;
    }
    __syncthreads();
    acc = *reinterpret_cast<uint4*>(smem + lane * 16 + (wave % 8) * 1024);
    if (acc.x == 0x12345678u) sink[0] = acc.y;
    (void)nw;
}

template <int D, int MODE>
void run(const uint32_t* src, unsigned bytes, uint32_t* sink, int waves) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<D, MODE>), dim3(256), dim3(waves * 64), 0, 0, src, bytes, 50, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<D, MODE>), dim3(256), dim3(waves * 64), 0, 0, src, bytes, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_cu = (double)iters * D * 1024 * waves;
    printf("mode=%s waves/CU=%2d depth=%2d : %7.1f B/ns/CU (%.1f B/clk @2.4GHz)  %.0f ns per wave-iteration\n", MODE ? "vgpr+ds_write" : "lds-dma      ",
           waves, D, bytes_cu / (ms * 1e6), bytes_cu / (ms * 1e6) / 2.4, ms * 1e6 / iters);
}

int main() {
    const unsigned bytes = 32u << 20;     // 32 MB window (L2 / MALL resident after warm-up)
    uint32_t *src, *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 64);
    hipMemset(src, 1, bytes);
    for (int waves : {4, 8, 16}) {
        run<4, 0>(src, bytes, sink, waves); run<8, 0>(src, bytes, sink, waves); run<16, 0>(src, bytes, sink, waves); run<32, 0>(src, bytes, sink, waves);
        run<4, 1>(src, bytes, sink, waves); run<8, 1>(src, bytes, sink, waves); run<16, 1>(src, bytes, sink, waves);
    }
    return 0;
}
