// Functional probe of global_load_lds_dwordx4: lane i of a wave lands at lds_base + 16*i; per-lane
// global source; out-of-range lanes redirected to a zero page.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ float g_zero[64];
__global__ void k(const float* __restrict__ g, float* out, int nvalid) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = wave * 8 + (lane >> 3), c4 = lane & 7;          // 8 rows of 128 B per wave-instruction
    const float* src = (row < nvalid) ? g + ((size_t)blockIdx.x * 32 + (31 - row)) * 32 + c4 * 4 : g_zero + c4 * 4;   // reversed rows
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + wave * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[(size_t)blockIdx.x * 1024 + i] = smem[i];
}
int main() {
    const int nb = 64; std::vector<float> h(nb * 1024), o(nb * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *d, *dout; hipMalloc(&d, h.size() * 4); hipMalloc(&dout, o.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<nb, 256, 4096>>>(d, dout, 29);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < nb; ++b) for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        float want = r < 29 ? h[((size_t)b * 32 + (31 - r)) * 32 + c] : 0.f;
        if (o[(size_t)b * 1024 + r * 32 + c] != want) ++bad;
    }
    printf("dma test: %d mismatches\n", bad);
    return bad != 0;
}
