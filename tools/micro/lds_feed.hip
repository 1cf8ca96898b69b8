// Ceiling of an LDS-fed fp32 MFMA loop on gfx950: 8 waves per CU (two per SIMD), every MFMA takes fresh A and B operands from LDS (conflict-free 8-byte reads, as the
// Winograd kernels issue them), nothing else in the loop.  16x16x4 needs two dwords per lane per 2048 FLOP, 32x32x2 the same two dwords per 4096 FLOP.
//   hipcc -O3 -w --offload-arch=gfx950 -o tools/micro/lds_feed tools/micro/lds_feed.hip && tools/micro/lds_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 ld2(const float* p) { return *(const volatile __attribute__((address_space(3))) f32x2*)p; }

// MODE 0: 16x16x4, 36 accumulators of 4 (the Winograd kernels' wave tile), operands a b64 pair per two MFMAs; MODE 1: 32x32x2, 9 accumulators of 16, a b64 pair per two MFMAs
// FEED 0: operands from registers (no LDS), 1: A from LDS, 2: A and B from LDS
// NV: independent fp32 FMAs per MFMA in the same wave (the input transform's share of the issue slots)
template <int MODE, int FEED, int NV = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 16384; i += 512) smem[i] = (i % 13) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* A = smem + wave * 1024 + 2 * lane;                  // lane-consecutive 8-byte pieces: conflict-free
    const float* B = smem + 8192 + wave * 1024 + 2 * lane;
    float s = 0.f;
    float fz[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const float fm = 1.0001f + 1e-6f * lane;
    if (MODE == 0) {
        f32x4 acc[36];
        for (int a = 0; a < 36; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x2 ar = {1.f, 2.f}, br = {3.f, 4.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int pp = 0; pp < 18; ++pp) {
                const f32x2 a = FEED >= 1 ? ld2(A + 128 * (pp & 7)) : ar;
                const f32x2 b = FEED >= 2 ? ld2(B + 128 * (pp & 7)) : br;
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[2 * pp], 0, 0, 0);
                acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[2 * pp + 1], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 2 * NV; ++v) fz[v & 7] = fmaf(fz[v & 7], fm, 0.5f);
            }
        }
        for (int a = 0; a < 36; ++a) s += acc[a][0] + acc[a][3];
        for (int v = 0; v < 8; ++v) s += fz[v];
    } else {
        f32x16 acc[9];
        for (int a = 0; a < 9; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        f32x2 ar = {1.f, 2.f}, br = {3.f, 4.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int pp = 0; pp < 9; ++pp) {                          // nine points x two k-steps (a 4-channel chunk): 18 MFMAs of 64 cycles = the same 1152 cycles
                const f32x2 a = FEED >= 1 ? ld2(A + 128 * (pp & 7)) : ar;
                const f32x2 b = FEED >= 2 ? ld2(B + 128 * (pp & 7)) : br;
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[pp], 0, 0, 0);
                acc[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[pp], 0, 0, 0);
            }
        }
        for (int a = 0; a < 9; ++a) s += acc[a][0] + acc[a][15];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE, int FEED, int NV = 0> void run(float* d, const char* what) {
    const int iters = 400, grid = 256;
    hipFuncSetAttribute((const void*)k<MODE, FEED, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, FEED, NV><<<grid, 512, 65536>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<MODE, FEED, NV><<<grid, 512, 65536>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double flop = (double)grid * 8 * iters * (MODE == 0 ? 36 * 2048.0 : 18 * 4096.0);
    printf("%-60s %.3f ms  %6.1f TFLOP/s (%.0f %% of 157.3)\n", what, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 0>(d, "16x16x4, operands in registers");
    run<0, 1>(d, "16x16x4, A from LDS (ds_read_b64 per two MFMAs)");
    run<0, 2>(d, "16x16x4, A and B from LDS");
    run<0, 2, 2>(d, "16x16x4, A and B from LDS + 2 VALU per MFMA");
    run<0, 2, 4>(d, "16x16x4, A and B from LDS + 4 VALU per MFMA");
    run<0, 2, 6>(d, "16x16x4, A and B from LDS + 6 VALU per MFMA");
    run<0, 0, 4>(d, "16x16x4, operands in registers + 4 VALU per MFMA");
    run<1, 0>(d, "32x32x2, operands in registers");
    run<1, 1>(d, "32x32x2, A from LDS");
    run<1, 2>(d, "32x32x2, A and B from LDS");
    return 0;
}
