// mrdis_wino4.h -- shared between mrdis_wino4.hip (the F(4x4, 3x3) convolution kernel) and mrdis_wino2.hip (the filter-image launch).
//
// Filter image of format 4: U = G g G^T with the 6 x 3 matrix G of F(4x4, 3x3) (interpolation points 0, +-1, +-2, inf), 36 values
// per (reduction channel, cout) pair, laid out as the LDS image that wino4_kernel copies by LDS-DMA, one 36 KB piece per
// (64-cout tile, 4-channel chunk):
//     [cout tile][chunk][18 point pairs][4 channels kq][128 slots],  slot of (cout m, point parity) = (2 m + parity + 32 kq) & 127
// (the two points of a pair sit next to each other: one ds_read_b64 is the A operand of two MFMAs; the rotation by 32 kq spreads
// the four k-rows of an MFMA operand over all 64 banks).  Zero where the chunk / tile runs past R / S.
#pragma once
#include <hip/hip_runtime.h>

#define MRDIS_W4_KC 4                    // reduction channels per chunk (one MFMA k-step of v_mfma_f32_16x16x4_f32)
#define MRDIS_W4_UPP 512                 // floats per point pair of the U image (4 channels x 64 couts x 2 points)
#define MRDIS_W4_UCHUNK (18 * MRDIS_W4_UPP)

// which image format (and kernel) a 3x3 stride-1 filter with R reduction channels and S couts gets: 4 = F(4x4, 3x3) (mrdis_wino4.hip),
// 2 = F(2x2, 3x3) (mrdis_wino2.hip).  A function of the filter alone -- the image is built once per step, before any call's shape is known;
// a call whose shape the F(4x4) kernel declines runs the F(2x2) kernel with its in-kernel filter transform.
int mrdis_wino_u_fmt(int R, int S, int spadeC);
int mrdis_wino_u_fmt_at(int R, int S, int spadeC, int level);
bool mrdis_wino_u_fmt_valid(int R, int S, int spadeC, int fmt);

// floats of the 36-point part of a format-4 image (the 16-point image of the same filter follows it: the fallback for calls the F(4x4) kernel declines)
static inline __host__ __device__ long long mrdis_wino4_image_floats(int R, int S, int spadeC) {
    const int tiles = spadeC ? (spadeC + 31) / 32 : (S + 63) / 64;
    return (long long)tiles * ((R + MRDIS_W4_KC - 1) / MRDIS_W4_KC) * MRDIS_W4_UCHUNK;
}
// format 5 (the narrow form, <= 32 couts): [32-cout tile][chunk of 4][18 point pairs][4 kq][64 slots], slot of (cout m, parity) = (2 m + parity + 32 kq) & 63;
// no 16-point image behind it (the F(2x2) kernel for 32 couts has no image path)
#define MRDIS_W4N_UCHUNK (18 * 256)
static inline __host__ __device__ long long mrdis_wino4n_image_floats(int R, int S) {
    return (long long)((S + 31) / 32) * ((R + MRDIS_W4_KC - 1) / MRDIS_W4_KC) * MRDIS_W4N_UCHUNK;
}

#ifdef __HIPCC__
// the 36 values of one (reduction channel, cout) pair from its nine taps g[3 * row + col]
__device__ __forceinline__ void mrdis_w4_filter_transform(const float g[9], float U[36]) {
    float t[6][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {                  // G g: G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
        const float g0 = g[q], g1 = g[3 + q], g2 = g[6 + q];
        const float s02 = g0 + g2;
        const float e = 0.041666667f * g0 + 0.16666667f * g2;
        t[0][q] = 0.25f * g0;
        t[1][q] = -0.16666667f * (s02 + g1);
        t[2][q] = -0.16666667f * (s02 - g1);
        t[3][q] = e + 0.083333333f * g1;
        t[4][q] = e - 0.083333333f * g1;
        t[5][q] = g2;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {                  // (G g) G^T
        const float g0 = t[a][0], g1 = t[a][1], g2 = t[a][2];
        const float s02 = g0 + g2;
        const float e = 0.041666667f * g0 + 0.16666667f * g2;
        U[6 * a + 0] = 0.25f * g0;
        U[6 * a + 1] = -0.16666667f * (s02 + g1);
        U[6 * a + 2] = -0.16666667f * (s02 - g1);
        U[6 * a + 3] = e + 0.083333333f * g1;
        U[6 * a + 4] = e - 0.083333333f * g1;
        U[6 * a + 5] = g2;
    }
}
#endif
