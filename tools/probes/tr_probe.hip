// Probe (GPU): semantics of ds_read_b64_tr_b16 and the A/B k-mapping of v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16.
// build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe ; run: ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ void tr_kernel(short* out, int mode) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    int l = threadIdx.x;
    int addr = (mode == 0) ? l * 4 : ((l & 15) / 4 * 16 + (l & 3) * 4 + (l >> 4) * 64);   // element units
    short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_ __attribute__((address_space(3)))*)(lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

// A[i][k] = (i==ai && k==ak), B[k][j] = (k==bk) -> D[ai][j] = (ak==bk).  Each lane loads element e of its 8 as k = kmap(lane, e).
__global__ void mfma16_kernel(float* out, int ak) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        int k = 8 * (l >> 4) + e;                          // hypothesis: lane group g holds k = 8g..8g+7
        a[e] = (__bf16)(((l & 15) == 3 && k == ak) ? 1.0f : 0.0f);
        b[e] = (__bf16)((float)(k + 1));                  // B[k][j] = k+1 for all j
    }
    floatx4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
__global__ void mfma32_kernel(float* out, int ak) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        int k = 8 * (l >> 5) + e;                          // hypothesis: half h holds k = 8h..8h+7
        a[e] = (__bf16)(((l & 31) == 5 && k == ak) ? 1.0f : 0.0f);
        b[e] = (__bf16)((float)(k + 1) + 100.0f * (l & 31));   // B[k][j] = k+1 + 100*j
    }
    floatx16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = acc[r];
}

int main() {
    short* d; hipMalloc(&d, 256 * 2);
    for (int mode = 0; mode < 2; ++mode) {
        tr_kernel<<<1, 64>>>(d, mode);
        std::vector<short> h(256); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("tr mode %d (lane: 4 received element indices)\n", mode);
        for (int l = 0; l < 64; ++l) printf("%2d:[%3d %3d %3d %3d]%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 4 == 3) ? "\n" : "  ");
    }
    float* f; hipMalloc(&f, 64 * 16 * 4);
    for (int ak : {0, 5, 13, 31}) {
        mfma16_kernel<<<1, 64>>>(f, ak);
        std::vector<float> h(256); hipMemcpy(h.data(), f, 1024, hipMemcpyDeviceToHost);
        // D[3][j]: row 3 -> lane group 0 (rows 0-3), reg 3; col j = lane & 15
        printf("mfma16 ak=%d: D[3][0]=%g (expect %d)  D[2][0]=%g\n", ak, h[0 * 4 + 3], ak + 1, h[0 * 4 + 2]);
    }
    for (int ak : {0, 7, 9, 15}) {
        mfma32_kernel<<<1, 64>>>(f, ak);
        std::vector<float> h(1024); hipMemcpy(h.data(), f, 4096, hipMemcpyDeviceToHost);
        // D[5][j]: row 5 = (reg&3)+8*(reg>>2)+4*(lane>>5): reg=1, lane>>5=1 -> lanes 32..63, reg 1 ; col j = lane&31
        printf("mfma32 ak=%d: D[5][2]=%g (expect %d)\n", ak, h[(32 + 2) * 16 + 1], ak + 1 + 200);
    }
    return 0;
}
