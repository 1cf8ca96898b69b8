// Sustained-rate probe for v_mfma_f32_32x32x2_f32 on gfx950: NW waves per SIMD issue
// back-to-back MFMAs on NACC independent accumulators, no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f + 1.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, float* d) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<NACC><<<grid, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flop = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d blocks/CU=%d : %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flop / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1>(1, d); run<2>(1, d); run<4>(1, d); run<1>(2, d); run<2>(2, d); run<2>(3, d); run<4>(2, d);
    return 0;
}
