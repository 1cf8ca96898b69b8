// How the shape of a wave's store instructions changes the write rate of an NHWC bf16 tile epilogue on gfx950.
// Every pattern writes the same bytes: 128 B per pixel (64 bf16 channels), 32 consecutive pixels per wave-block.
//   A: what a 32x32x16 MFMA accumulator gives directly: 8 B per lane, the two half-waves side by side -> 16 B per 128-B line per instruction, 8 instructions
//   B: after a v_permlane32_swap of packed pairs: 16 B per lane -> 32 B per line per instruction, 4 instructions
//   C: after a transpose (LDS): 16 B per lane, 8 lanes per line -> whole 128-B lines, 4 instructions
//   D: 16 B per lane, 4 lanes per 64-B half line (two instructions complete a line)
// hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(512) void k(char* out, long n_blocks) {   // block = 32 px x 128 B = 4 KB, one per wave per trip
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 31, half = lane >> 5;
    for (long b = (long)blockIdx.x * 8 + wave; b < n_blocks; b += (long)gridDim.x * 8) {
        char* base = out + b * 4096;
        if (PAT == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 v = {(unsigned)b + g, (unsigned)lane + t};
                    u32x2* p = (u32x2*)(base + px * 128 + 64 * t + 16 * g + 8 * half);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
        } else if (PAT == 1) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 v = {(unsigned)b + h, (unsigned)lane + t, 3u, 4u};
                    u32x4* p = (u32x4*)(base + px * 128 + 64 * t + 32 * h + 16 * half);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
        } else if (PAT == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4 v = {(unsigned)b + i, (unsigned)lane, 3u, 4u};
                u32x4* p = (u32x4*)(base + i * 1024 + lane * 16);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u32x4 v = {(unsigned)b + i, (unsigned)lane + t, 3u, 4u};
                    u32x4* p = (u32x4*)(base + (16 * i + (lane >> 2)) * 128 + 64 * t + 16 * (lane & 3));
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
        }
    }
}

template <int PAT, bool NT> void run(char* d, long bytes_per_buf, int nbuf, int wg_per_cu, const char* name) {
    const long n_blocks = bytes_per_buf / 4096;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < nbuf; ++r) k<PAT, NT><<<grid, 512>>>(d + r * bytes_per_buf, n_blocks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 3 * nbuf;
    for (int r = 0; r < reps; ++r) k<PAT, NT><<<grid, 512>>>(d + (r % nbuf) * bytes_per_buf, n_blocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-58s %s  %d WG/CU: %7.1f us  %6.0f GB/s\n", name, NT ? "nt" : "  ", wg_per_cu, ms * 1e3, bytes_per_buf / ms / 1e6);
}

int main() {
    const long bytes = 32L * 256 * 256 * 128;   // the sp6 gamma|beta output, 268 MB
    const int nbuf = 4;
    char* d; if (hipMalloc(&d, bytes * nbuf) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int wg = 1; wg <= 4; wg *= 2) {
        run<0, false>(d, bytes, nbuf, wg, "A  8 B/lane, 16 B per line per instruction (MFMA layout)");
        run<1, false>(d, bytes, nbuf, wg, "B 16 B/lane, 32 B per line per instruction (permlane32_swap)");
        run<3, false>(d, bytes, nbuf, wg, "D 16 B/lane, 64 B per line per instruction");
        run<2, false>(d, bytes, nbuf, wg, "C 16 B/lane, whole lines (transpose)");
    }
    run<0, true>(d, bytes, nbuf, 1, "A  8 B/lane, 16 B per line per instruction (MFMA layout)");
    run<1, true>(d, bytes, nbuf, 1, "B 16 B/lane, 32 B per line per instruction (permlane32_swap)");
    run<3, true>(d, bytes, nbuf, 1, "D 16 B/lane, 64 B per line per instruction");
    run<2, true>(d, bytes, nbuf, 1, "C 16 B/lane, whole lines (transpose)");
    hipFree(d);
    return 0;
}
