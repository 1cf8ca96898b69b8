// Probe: MFMA utilisation of the tapconv inner loop shape (LDS operand reads + f32 MFMA),
// no global traffic.  Knobs: MSUB x NSUB accumulators per wave, blocks per CU via LDS size.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MSUB, int NSUB, int KC, int BN, int PIPE>
__global__ __launch_bounds__(256) void k(float* out, int reps, int ntaps) {
    extern __shared__ float smem[];
    constexpr int S = KC + 1;
    float* ws = smem;                     // [16][KC][BN]
    float* xs = smem + 9 * KC * BN;       // [204][S]
    for (int i = threadIdx.x; i < 9 * KC * BN + 204 * S; i += 256) smem[i] = (i % 13) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WN = (BN / 32) / NSUB, WM = 4 / WN;
    const int wave_m = wave / WN, wave_n = wave % WN;
    int abase[MSUB];
    for (int i = 0; i < MSUB; ++i) abase[i] = ((((wave_m * MSUB + i) & 3) + 1) * 34 + 1 + (lane & 31)) * S + (lane >> 5);
    const int bbase = (lane >> 5) * BN + wave_n * NSUB * 32 + (lane & 31);
    f32x16 acc[MSUB][NSUB];
    for (int i = 0; i < MSUB; ++i) for (int j = 0; j < NSUB; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (PIPE == 2) {
        constexpr int G = (MSUB * NSUB >= 4) ? 1 : (MSUB * NSUB == 2 ? 2 : 4), NG = (KC / 2) / G;
        float an[G][MSUB], bn[G][NSUB];
        for (int s_ = 0; s_ < G; ++s_) { for (int i = 0; i < MSUB; ++i) an[s_][i] = xs[abase[i] + 2 * s_]; for (int j = 0; j < NSUB; ++j) bn[s_][j] = ws[bbase + 2 * s_ * BN + j * 32]; }
        for (int rep = 0; rep < reps; ++rep)
            for (int t = 0; t < ntaps; ++t) {
                const int toff = ((t / 3 - 1) * 34 + (t % 3 - 1)) * S;
                const int tn = (t + 1) % ntaps;
                const int toffn = ((tn / 3 - 1) * 34 + (tn % 3 - 1)) * S;
                const float* wt = ws + t * (KC * BN) + bbase;
                const float* wtn = ws + tn * (KC * BN) + bbase;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float av[G][MSUB], bv[G][NSUB];
#pragma unroll
                    for (int s_ = 0; s_ < G; ++s_) { for (int i = 0; i < MSUB; ++i) av[s_][i] = an[s_][i]; for (int j = 0; j < NSUB; ++j) bv[s_][j] = bn[s_][j]; }
#pragma unroll
                    for (int s_ = 0; s_ < G; ++s_) {
                        if (g + 1 < NG) { const int kk = (g + 1) * G + s_;
                            for (int i = 0; i < MSUB; ++i) an[s_][i] = xs[abase[i] + toff + 2 * kk];
                            for (int j = 0; j < NSUB; ++j) bn[s_][j] = wt[2 * kk * BN + j * 32];
                        } else {
                            for (int i = 0; i < MSUB; ++i) an[s_][i] = xs[abase[i] + toffn + 2 * s_];
                            for (int j = 0; j < NSUB; ++j) bn[s_][j] = wtn[2 * s_ * BN + j * 32];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s_ = 0; s_ < G; ++s_)
#pragma unroll
                        for (int i = 0; i < MSUB; ++i)
#pragma unroll
                            for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s_][i], bv[s_][j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    } else
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < ntaps; ++t) {
            const int toff = ((t / 3 - 1) * 34 + (t % 3 - 1)) * S;
            const float* wt = ws + t * (KC * BN) + bbase;
#pragma unroll
            for (int kk = 0; kk < KC / 2; ++kk) {
                float av[MSUB], bv[NSUB];
#pragma unroll
                for (int i = 0; i < MSUB; ++i) av[i] = xs[abase[i] + toff + 2 * kk];
#pragma unroll
                for (int j = 0; j < NSUB; ++j) bv[j] = wt[2 * kk * BN + j * 32];
                if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MSUB; ++i)
#pragma unroll
                    for (int j = 0; j < NSUB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < MSUB; ++i) for (int j = 0; j < NSUB; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MSUB, int NSUB, int KC, int BN, int PIPE> void run(int bpc, float* d, const char* name) {
    const int reps = 40, ntaps = 9, grid = 256 * bpc;
    size_t lds = (9 * KC * BN + 204 * (KC + 1)) * 4;
    size_t want = (160 * 1024 / bpc) & ~255u; if (lds < want - 2048) lds = want - 2048;   // exactly bpc blocks per CU
    hipFuncSetAttribute((const void*)k<MSUB, NSUB, KC, BN, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MSUB, NSUB, KC, BN, PIPE><<<grid, 256, lds>>>(d, reps, ntaps); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k<MSUB, NSUB, KC, BN, PIPE><<<grid, 256, lds>>>(d, reps, ntaps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double flop = (double)grid * 4 * reps * ntaps * (KC / 2) * MSUB * NSUB * 4096.0;
    printf("%-28s blocks/CU=%d lds=%zuKB : %.3f ms  %.1f TF/s\n", name, bpc, lds / 1024, ms, flop / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int b = 1; b <= 3; ++b) run<2, 1, 16, 64, 0>(b, d, "BN=64 naive");
    for (int b = 1; b <= 3; ++b) run<2, 1, 16, 64, 2>(b, d, "BN=64 pipelined");
    for (int b = 1; b <= 4; b += 1) run<1, 1, 16, 32, 0>(b, d, "BN=32 naive");
    for (int b = 1; b <= 4; b += 1) run<1, 1, 16, 32, 2>(b, d, "BN=32 pipelined");
    for (int b = 1; b <= 3; ++b) run<2, 2, 8, 128, 0>(b, d, "BN=128 naive");
    for (int b = 1; b <= 3; ++b) run<2, 2, 8, 128, 2>(b, d, "BN=128 pipelined");
    return 0;
}
