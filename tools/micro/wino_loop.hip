// Probe: the MFMA phase of wino2_kernel in isolation (no global traffic, no transforms): per chunk 16 Winograd points x
// (3 ds_read_b64 + 4 v_mfma_f32_16x16x4_f32), 128 accumulators per wave, 8 waves per CU (2 per SIMD), one workgroup per CU.
// MODE 0: MFMAs only (operands in registers)   1: + LDS operand reads two steps ahead + one barrier per chunk (wino2's loop)
// MODE 2: as 1 without the barrier             3: as 1, operand reads without the sched_group_barrier interleave
// MODE 4: as 1 with 2 dependent-free accumulators order (acc0 x, acc1 x, acc1 y, acc0 y)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int XI = 512, UVBUF = 16 * XI;
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void k(float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 4 * UVBUF; i += NT) smem[i] = (i % 13) * 0.01f;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int cg = wave & 1, tg = (wave >> 1) & 3;
    int a0_off = kq * 128 + ((2 * (32 * cg + l16) + 32 * kq) & 127);
    int a1_off = kq * 128 + ((2 * (32 * cg + 16 + l16) + 32 * kq) & 127);
    int b_off = kq * 128 + ((2 * (16 * tg + l16) + 32 * kq) & 127);
    f32x4 acc[16][2];
    for (int s = 0; s < 16; ++s) for (int h = 0; h < 2; ++h) acc[s][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    float2 r0 = make_float2(tid * 1e-3f, 1.f), r1 = make_float2(0.5f, tid * 2e-3f), r2 = make_float2(0.25f, 0.125f);
    for (int c = 0; c < chunks; ++c) {
        const int P = c & 1;
        int ia0 = P * UVBUF + a0_off, ia1 = P * UVBUF + a1_off, ib = (2 + P) * UVBUF + b_off;
        asm volatile("" : "+v"(ia0), "+v"(ia1), "+v"(ib));
        const float* Ua0 = smem + ia0; const float* Ua1 = smem + ia1; const float* Vbv = smem + ib;
        float2 a0[3], a1[3], bv[3];
        if (MODE == 0) { for (int s = 0; s < 3; ++s) { a0[s] = r0; a1[s] = r1; bv[s] = r2; } }
        else {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                a0[s_] = *reinterpret_cast<const float2*>(Ua0 + s_ * XI);
                a1[s_] = *reinterpret_cast<const float2*>(Ua1 + s_ * XI);
                bv[s_] = *reinterpret_cast<const float2*>(Vbv + s_ * XI);
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            if (MODE != 0 && s_ + 2 < 16) {
                a0[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Ua0 + (s_ + 2) * XI);
                a1[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Ua1 + (s_ + 2) * XI);
                bv[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Vbv + (s_ + 2) * XI);
            }
            const int c_ = s_ % 3;
            if (MODE == 4) {
                acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c_].x, bv[c_].x, acc[s_][0], 0, 0, 0);
                acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c_].x, bv[c_].x, acc[s_][1], 0, 0, 0);
                acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c_].y, bv[c_].y, acc[s_][1], 0, 0, 0);
                acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c_].y, bv[c_].y, acc[s_][0], 0, 0, 0);
            } else {
                acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c_].x, bv[c_].x, acc[s_][0], 0, 0, 0);
                acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c_].x, bv[c_].x, acc[s_][1], 0, 0, 0);
                acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c_].y, bv[c_].y, acc[s_][0], 0, 0, 0);
                acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c_].y, bv[c_].y, acc[s_][1], 0, 0, 0);
            }
            if (MODE != 3) {
#pragma unroll
                for (int g_ = 0; g_ < 4; ++g_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1 || MODE == 3 || MODE == 4) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < 16; ++a) for (int h = 0; h < 2; ++h) for (int r = 0; r < 4; ++r) s += acc[a][h][r];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int MODE, int NT> void run(float* d, const char* what) {
    const int chunks = 4000, grid = 256;
    const size_t lds = 4 * UVBUF * sizeof(float) + 24 * 1024;      // 152 KB: one workgroup per CU
    hipFuncSetAttribute((const void*)k<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NT><<<grid, NT, lds>>>(d, chunks); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<MODE, NT><<<grid, NT, lds>>>(d, chunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = (double)grid * (NT / 64) * chunks * 64 * 2048.0;
    printf("mode %d, %d waves/CU (%s): %.3f ms  %.1f TFLOP/s  (%.0f %% of 157.3)\n", MODE, NT / 64, what, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 512>(d, "MFMA only");
    run<1, 512>(d, "wino2 loop: reads 2 ahead + barrier per chunk");
    run<2, 512>(d, "no barrier");
    run<3, 512>(d, "no sched_group interleave");
    run<4, 512>(d, "acc order 0 1 1 0");
    run<0, 256>(d, "MFMA only, 1 wave/SIMD");
    run<2, 256>(d, "reads, no barrier, 1 wave/SIMD");
    return 0;
}
