// Probe: does a 16-byte buffer_load ... lds (LDS-DMA) zero-fill the LDS slot of a lane whose offset fails the
// buffer range check?  And where do lanes land (wave-uniform base + lane*16)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t* src, unsigned src_bytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2 * 64 * 4; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, src_bytes, 0x00020000);
    // lanes 0..63: lane%5==0 -> out of range, else permuted source (63-lane)
    unsigned voff = (lane % 5 == 0) ? 0xFFFFFFFFu : (unsigned)(63 - lane) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 64 * 4), 16, voff, 0, 0, 0);   // second KB of the array
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2 * 64 * 4; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<uint32_t> h(64 * 4);
    for (int i = 0; i < 64 * 4; ++i) h[i] = 1000 + i;
    uint32_t *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 2 * 64 * 4 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, (unsigned)(h.size() * 4), o);
    std::vector<uint32_t> r(2 * 64 * 4);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64 * 4; ++i) if (r[i] != 0xABABABABu) ++bad;
    printf("first KB untouched: %s\n", bad ? "NO" : "yes");
    for (int lane = 0; lane < 12; ++lane)
        printf("lane %2d: %08x %08x %08x %08x  (expect %s)\n", lane, r[256 + lane * 4], r[256 + lane * 4 + 1], r[256 + lane * 4 + 2],
               r[256 + lane * 4 + 3], lane % 5 == 0 ? "zeros if OOB zero-fills" : "src of lane 63-l");
    int ok = 1;
    for (int lane = 0; lane < 64; ++lane)
        for (int k = 0; k < 4; ++k) {
            uint32_t want = lane % 5 == 0 ? 0u : 1000 + (63 - lane) * 4 + k;
            if (r[256 + lane * 4 + k] != want) ok = 0;
        }
    printf("RESULT %s\n", ok ? "OOB_ZERO_FILL_AND_LANE_LINEAR_OK" : "MISMATCH");
    return 0;
}
