// mrdis_c16.hip -- 3x3 / stride 1 / pad 1 convolution with exactly 16 output channels and 16 or 32 input channels, fp32
// (sp6.out 32 -> 16 at full resolution, model.py:2446: 16 calls per step, 19.3 GFLOP each at B = 32).  The generic narrow-output kernel
// (tapconv16_kernel) keeps the matrix pipe 46 % busy there: operands are single-float LDS reads issued right before the MFMA that needs
// them, two barriers per 16-channel chunk, 128-position tiles.  Here
//   * the whole filter lives in registers as the A operand of v_mfma_f32_16x16x4_f32 (9 taps x Ci / 4 k-steps = 72 VGPRs for Ci = 32):
//     the only LDS traffic of the MFMA phase is the pixel operand;
//   * a k-step's four channels are (4 kq + j) for lane group kq, so ONE ds_read_b128 at pixel * pitch + 16 h + 4 kq feeds four k-steps
//     (pitch = Ci + 4 floats: the 16 pixels of a tile land on 16 distinct 4-bank groups);
//   * a wave owns four 16-pixel tiles (two rows x 32 columns of an 8 x 32 output tile) that share every A register: 16 MFMAs per four
//     reads, the reads of the next (tap, half) in flight while they run;
//   * workgroups are persistent (grid stride over 8 x 32 tiles) with the next tile's halo'd input block in flight in registers during
//     the 288 MFMAs of the current one; 49 KB of LDS, two workgroups per CU fill each other's staging phases.
// D[cout][pixel]: lane = pixel, 4 registers = 4 consecutive couts -> 16-byte stores, 64 contiguous bytes per pixel.
#include "mrdis_common.h"

struct C16Params {
    const float* x; const float* w; const float* bias; float* y;
    int N, H, W, Ci, ldx, ldy, lrelu;
    int tilesY, tilesX, ntiles;
    unsigned x_bytes;
};

namespace {
constexpr int C16_TH = 8, C16_TW = 32, C16_RW = C16_TW + 2, C16_NPX = (C16_TH + 2) * C16_RW;     // 340 pixels of input per tile
constexpr unsigned C16_OOB = 0xfffffff0u;
typedef unsigned c16_u32x4 __attribute__((ext_vector_type(4)));
}

template <int HALVES>
__global__ __launch_bounds__(256, 2) void conv3x3_c16_kernel(const C16Params p) {
    constexpr int CI = 16 * HALVES, PP = CI + 4, Q = CI / 4;         // LDS pixel pitch (floats), float4 pieces per pixel
    constexpr int XR = (C16_NPX * Q + 255) / 256;                    // staging items per thread (11 for Ci = 32)
    constexpr int NG = 9 * HALVES;                                   // (tap, half) groups: 4 k-steps each
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [340][PP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;

    // the filter: a[g][j] = w[tap][16 h + 4 kq + j][l16]
    float a[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[g][j] = p.w[((g / HALVES) * CI + 16 * (g % HALVES) + 4 * kq + j) * 16 + l16];
    const float4 bv = p.bias ? make_float4(p.bias[4 * kq], p.bias[4 * kq + 1], p.bias[4 * kq + 2], p.bias[4 * kq + 3])      // (a parameter view: 4-byte aligned)
                              : make_float4(0.f, 0.f, 0.f, 0.f);

    // staging roles: item = (pixel of the 10 x 34 block, float4 piece)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    int s_l[XR], s_ry[XR], s_rx[XR], s_q[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + 256 * it, pi = idx / Q;
        s_q[it] = 4 * (idx - pi * Q);
        s_ry[it] = pi / C16_RW; s_rx[it] = pi - s_ry[it] * C16_RW;
        s_l[it] = idx < C16_NPX * Q ? pi * PP + s_q[it] : -1;
    }
    c16_u32x4 xr[XR];
    auto load_tile = [&](int tile) {                  // tile >= ntiles: every offset out of range -> zeros, no branch around a load
        const bool on = tile < p.ntiles;
        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int h = ty * C16_TH - 1 + s_ry[it], w_ = tx * C16_TW - 1 + s_rx[it];
            const bool ok = on && s_l[it] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldx + s_q[it]) : C16_OOB), 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it)
            if (s_l[it] >= 0) *reinterpret_cast<c16_u32x4*>(smem + s_l[it]) = xr[it];
    };

    // MFMA role: tiles t = 0..3 of the wave = (row 2 wave + (t >> 1), columns 16 (t & 1) ..)
    int boff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) boff[t] = ((2 * wave + (t >> 1)) * C16_RW + 16 * (t & 1) + l16) * PP + 4 * kq;

    int tile = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    load_tile(tile);
    store_tile();
    __syncthreads();
    for (; tile < p.ntiles; tile += gridDim.x) {
        load_tile(tile + gridDim.x);
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 b[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[0][t] = *reinterpret_cast<const float4*>(smem + boff[t]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
                const int tap = (g + 1) / HALVES, h = (g + 1) % HALVES;
                const int go = ((tap / 3) * C16_RW + (tap % 3)) * PP + 16 * h;
#pragma unroll
                for (int t = 0; t < 4; ++t) b[(g + 1) & 1][t] = *reinterpret_cast<const float4*>(smem + boff[t] + go);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 v = b[g & 1][t];
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][0], v.x, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][1], b[g & 1][t].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][2], b[g & 1][t].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][3], b[g & 1][t].w, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: bias, LeakyReLU, 16-byte stores (lane = pixel, couts 4 kq .. 4 kq + 3)
        {
            int t_ = tile;
            const int tx = t_ % p.tilesX; t_ /= p.tilesX;
            const int ty = t_ % p.tilesY;
            const int n = t_ / p.tilesY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int oy = ty * C16_TH + 2 * wave + (t >> 1), ox = tx * C16_TW + 16 * (t & 1) + l16;
                float4 o = make_float4(acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w);
                if (p.lrelu) { o.x = o.x > 0.f ? o.x : 0.2f * o.x; o.y = o.y > 0.f ? o.y : 0.2f * o.y; o.z = o.z > 0.f ? o.z : 0.2f * o.z; o.w = o.w > 0.f ? o.w : 0.2f * o.w; }
                if (oy < p.H && ox < p.W) *reinterpret_cast<float4*>(p.y + ((long long)(n * p.H + oy) * p.W + ox) * p.ldy + 4 * kq) = o;
            }
        }
        __syncthreads();
        store_tile();
        __syncthreads();
    }
}

// ---- six-product form (option split6; Ci = 32): the same convolution on the bf16 matrix pipe at fp32 accuracy.  The fp32 form above is bound by its 288
// fp32 MFMAs of 32 cycles per wave and tile (19.3 GFLOP at 113 TF/s = 0.72 of the fp32 MFMA peak; the layer moves 402 MB: 75 us at the rate the narrow
// layers stream).  Both operands become three bf16 terms, v = hi + mid + lo (each the bf16 rounding of what the terms before it left), and the six products of
// order <= 2 are summed in fp32 (dropped: < 2^-23 of a product): K = 32 channels is ONE v_mfma_f32_16x16x32_bf16, so a tap of a 16-pixel tile is 6 MFMAs of
// 16 cycles instead of 8 of 32.  The split of x happens once per input pixel, on its way from the staging registers into LDS: a pixel is three 64-byte runs
// [hi | mid | lo] of 32 bf16 (+ 16 bytes: pitch 208 B = 52 words, the 16 pixels of a tile land on 16 distinct 4-bank groups), and a lane's B operand of
// (tap, term) is one ds_read_b128.  The filter's three terms stay in registers (108 VGPRs).
typedef __bf16 c16_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 c16_bf16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void conv3x3_c16_split6_kernel(const C16Params p) {
    constexpr int CI = 32, Q = CI / 4, PB = 208;                     // float4 pieces per pixel, LDS pixel pitch in BYTES
    constexpr int XR = (C16_NPX * Q + 255) / 256;                    // staging items per thread (11)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];     // [340][208 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;

    // the filter: A operand of tap g, term: row l16 = cout, k-slot j <-> channel 8 kq + j
    c16_bf16x8 a[9][3];
#pragma unroll
    for (int g = 0; g < 9; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = p.w[(g * CI + 8 * kq + j) * 16 + l16];
            const __bf16 hi = (__bf16)v; const float r1 = v - (float)hi; const __bf16 mid = (__bf16)r1;
            a[g][0][j] = hi; a[g][1][j] = mid; a[g][2][j] = (__bf16)(r1 - (float)mid);
        }
    const float4 bv = p.bias ? make_float4(p.bias[4 * kq], p.bias[4 * kq + 1], p.bias[4 * kq + 2], p.bias[4 * kq + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    int s_l[XR], s_ry[XR], s_rx[XR], s_q[XR];                        // s_l: LDS byte offset of the item's hi quad, or -1
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + 256 * it, pi = idx / Q;
        s_q[it] = 4 * (idx - pi * Q);
        s_ry[it] = pi / C16_RW; s_rx[it] = pi - s_ry[it] * C16_RW;
        s_l[it] = idx < C16_NPX * Q ? pi * PB + 2 * s_q[it] : -1;
    }
    c16_u32x4 xr[XR];
    auto load_tile = [&](int tile) {                  // tile >= ntiles: every offset out of range -> zeros, no branch around a load
        const bool on = tile < p.ntiles;
        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int h = ty * C16_TH - 1 + s_ry[it], w_ = tx * C16_TW - 1 + s_rx[it];
            const bool ok = on && s_l[it] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldx + s_q[it]) : C16_OOB), 0, 0);
        }
    };
    auto store_tile = [&]() {                         // four fp32 channels -> their three bf16 terms, 8 bytes into each of the pixel's runs
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            if (s_l[it] < 0) continue;
            const float xv[4] = {__uint_as_float(xr[it].x), __uint_as_float(xr[it].y), __uint_as_float(xr[it].z), __uint_as_float(xr[it].w)};
            c16_bf16x4 hi, mid, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const __bf16 h_ = (__bf16)xv[c]; const float r1 = xv[c] - (float)h_; const __bf16 m_ = (__bf16)r1;
                hi[c] = h_; mid[c] = m_; lo[c] = (__bf16)(r1 - (float)m_);
            }
            *reinterpret_cast<c16_bf16x4*>(smem_b + s_l[it]) = hi;
            *reinterpret_cast<c16_bf16x4*>(smem_b + s_l[it] + 64) = mid;
            *reinterpret_cast<c16_bf16x4*>(smem_b + s_l[it] + 128) = lo;
        }
    };

    // MFMA role: tiles t = 0..3 of the wave = (row 2 wave + (t >> 1), columns 16 (t & 1) ..); B operand: pixel l16 of the tile, channels 8 kq .. + 7
    int boff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) boff[t] = ((2 * wave + (t >> 1)) * C16_RW + 16 * (t & 1) + l16) * PB + 16 * kq;

    int tile = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    load_tile(tile);
    store_tile();
    __syncthreads();
    for (; tile < p.ntiles; tile += gridDim.x) {
        load_tile(tile + gridDim.x);
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int go = ((g / 3) * C16_RW + (g % 3)) * PB;
            c16_bf16x8 xh[4], xm[4], xl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xh[t] = *reinterpret_cast<const c16_bf16x8*>(smem_b + boff[t] + go);
                xm[t] = *reinterpret_cast<const c16_bf16x8*>(smem_b + boff[t] + go + 64);
                xl[t] = *reinterpret_cast<const c16_bf16x8*>(smem_b + boff[t] + go + 128);
            }
            // smallest products first; the four tiles of a wave share every A register
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][1], xm[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][0], xl[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][2], xh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][0], xm[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][1], xh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g][0], xh[t], acc[t], 0, 0, 0);
        }
        {
            int t_ = tile;
            const int tx = t_ % p.tilesX; t_ /= p.tilesX;
            const int ty = t_ % p.tilesY;
            const int n = t_ / p.tilesY;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int oy = ty * C16_TH + 2 * wave + (t >> 1), ox = tx * C16_TW + 16 * (t & 1) + l16;
                float4 o = make_float4(acc[t][0] + bv.x, acc[t][1] + bv.y, acc[t][2] + bv.z, acc[t][3] + bv.w);
                if (p.lrelu) { o.x = o.x > 0.f ? o.x : 0.2f * o.x; o.y = o.y > 0.f ? o.y : 0.2f * o.y; o.z = o.z > 0.f ? o.z : 0.2f * o.z; o.w = o.w > 0.f ? o.w : 0.2f * o.w; }
                if (oy < p.H && ox < p.W) *reinterpret_cast<float4*>(p.y + ((long long)(n * p.H + oy) * p.W + ox) * p.ldy + 4 * kq) = o;
            }
        }
        __syncthreads();
        store_tile();
        __syncthreads();
    }
}

// ---- the other direction of the same layer (option split6): 16 -> 32 channels, i.e. the DATA GRADIENT of sp6.out (dy 16 channels, dx 32; the filter is the
// layer's [tap][16][32] = w_tkc, taps reversed).  K = 16 channels is exactly one v_mfma_f32_32x32x16_bf16 (A = filter: 32 couts x 16 channels; B = 32 pixels
// of a row x 16 channels), six products per tap and 32-pixel tile.  Same structure as conv3x3_c16_split6_kernel: a pixel in LDS is [hi | mid | lo] runs of
// 16 bf16 (pitch 112 B), the filter's three terms stay in registers (108 VGPRs), a wave owns two rows of the 8 x 32 tile.  D[cout][pixel]: lane = pixel,
// registers = couts 8 g + 4 half .. + 3: four 16-byte stores per tile (32 bytes per 128-byte line and instruction).
__global__ __launch_bounds__(256, 2) void conv3x3_c16t_split6_kernel(const C16Params p) {
    constexpr int CI = 16, Q = CI / 4, PB = 112;                     // float4 pieces per pixel, LDS pixel pitch in bytes
    constexpr int XR = (C16_NPX * Q + 255) / 256;                    // staging items per thread (6)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];     // [340][112 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e = lane & 31, half = lane >> 5;

    // the filter: A operand of tap g, term: row e = cout (of 32), k-slot j <-> channel 8 half + j.  p.lrelu carries the tap flip (1: data gradient)
    c16_bf16x8 a[9][3];
#pragma unroll
    for (int g = 0; g < 9; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int gf = p.lrelu ? 8 - g : g;
            const float v = p.w[(gf * CI + 8 * half + j) * 32 + e];
            const __bf16 hi = (__bf16)v; const float r1 = v - (float)hi; const __bf16 mid = (__bf16)r1;
            a[g][0][j] = hi; a[g][1][j] = mid; a[g][2][j] = (__bf16)(r1 - (float)mid);
        }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    int s_l[XR], s_ry[XR], s_rx[XR], s_q[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + 256 * it, pi = idx / Q;
        s_q[it] = 4 * (idx - pi * Q);
        s_ry[it] = pi / C16_RW; s_rx[it] = pi - s_ry[it] * C16_RW;
        s_l[it] = idx < C16_NPX * Q ? pi * PB + 2 * s_q[it] : -1;
    }
    c16_u32x4 xr[XR];
    auto load_tile = [&](int tile) {
        const bool on = tile < p.ntiles;
        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int h = ty * C16_TH - 1 + s_ry[it], w_ = tx * C16_TW - 1 + s_rx[it];
            const bool ok = on && s_l[it] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldx + s_q[it]) : C16_OOB), 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            if (s_l[it] < 0) continue;
            const float xv[4] = {__uint_as_float(xr[it].x), __uint_as_float(xr[it].y), __uint_as_float(xr[it].z), __uint_as_float(xr[it].w)};
            c16_bf16x4 hi, mid, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const __bf16 h_ = (__bf16)xv[c]; const float r1 = xv[c] - (float)h_; const __bf16 m_ = (__bf16)r1;
                hi[c] = h_; mid[c] = m_; lo[c] = (__bf16)(r1 - (float)m_);
            }
            *reinterpret_cast<c16_bf16x4*>(smem_t + s_l[it]) = hi;
            *reinterpret_cast<c16_bf16x4*>(smem_t + s_l[it] + 32) = mid;
            *reinterpret_cast<c16_bf16x4*>(smem_t + s_l[it] + 64) = lo;
        }
    };
    // MFMA role: tiles t = 0, 1 of the wave = rows 2 wave + t, all 32 columns; B operand: pixel e of the row, channels 8 half .. + 7
    int boff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) boff[t] = ((2 * wave + t) * C16_RW + e) * PB + 16 * half;

    int tile = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    load_tile(tile);
    store_tile();
    __syncthreads();
    for (; tile < p.ntiles; tile += gridDim.x) {
        load_tile(tile + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int go = ((g / 3) * C16_RW + (g % 3)) * PB;
            c16_bf16x8 xh[2], xm[2], xl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xh[t] = *reinterpret_cast<const c16_bf16x8*>(smem_t + boff[t] + go);
                xm[t] = *reinterpret_cast<const c16_bf16x8*>(smem_t + boff[t] + go + 32);
                xl[t] = *reinterpret_cast<const c16_bf16x8*>(smem_t + boff[t] + go + 64);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][1], xm[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0], xl[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][2], xh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0], xm[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][1], xh[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][0], xh[t], acc[t], 0, 0, 0);
        }
        {
            int t_ = tile;
            const int tx = t_ % p.tilesX; t_ /= p.tilesX;
            const int ty = t_ % p.tilesY;
            const int n = t_ / p.tilesY;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int oy = ty * C16_TH + 2 * wave + t, ox = tx * C16_TW + e;
                if (oy < p.H && ox < p.W) {
                    float* dst = p.y + ((long long)(n * p.H + oy) * p.W + ox) * p.ldy + 4 * half;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4)
                        *reinterpret_cast<float4*>(dst + 8 * g4) = make_float4(acc[t][4 * g4], acc[t][4 * g4 + 1], acc[t][4 * g4 + 2], acc[t][4 * g4 + 3]);
                }
            }
        }
        __syncthreads();
        store_tile();
        __syncthreads();
    }
}

// the 16 -> 32 six-product kernel as a data gradient (dy (N, H, W, 16) -> dx (N, H, W, 32), w_tkc = [9][16][32], taps reversed) or a forward (flip = 0)
int mrdis_run_c16t_split6(const float* x, int ldx, const float* w_t_ci_co, float* y, int ldy, int N, int H, int W, int flip, hipStream_t s) {
    if (ldx % 4 != 0 || ldy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W < 65536 || mrdis_opt(MRDIS_OPT_NOW16)) return MRDIS_EUNSUPPORTED;
    // NOT part of the default policy (option split6 = 1): alone the kernel beats the layer's Winograd F(4x4) data gradient (149 vs 169 us), in the step it does
    // not (same-box A/B 133.1-133.5 vs 132.7-133.2 ms: the F(4x4) kernel moves the same bytes with a quarter of the multiplies).  split6 = 7 selects it.
    if (mrdis_opt(MRDIS_OPT_SPLIT6) != 7) return MRDIS_EUNSUPPORTED;
    const long long xb = 4LL * (((long long)N * H * W - 1) * ldx + 16);
    if (xb >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    C16Params p{};
    p.x = x; p.w = w_t_ci_co; p.bias = nullptr; p.y = y; p.N = N; p.H = H; p.W = W; p.Ci = 16; p.ldx = ldx; p.ldy = ldy; p.lrelu = flip;
    p.tilesY = mrdis_cdiv(H, C16_TH); p.tilesX = mrdis_cdiv(W, C16_TW);
    const long long nt = (long long)N * p.tilesY * p.tilesX;
    if (nt > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.ntiles = (int)nt; p.x_bytes = (unsigned)xb;
    const int grid = p.ntiles < 512 ? p.ntiles : 512;
    MRDIS_LAUNCH(conv3x3_c16t_split6_kernel, dim3(grid), dim3(256), (size_t)C16_NPX * 112, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// returns MRDIS_EUNSUPPORTED outside what the kernel covers (the caller then runs the generic narrow-output kernel)
int mrdis_run_c16(const float* x, int ldx, const float* w_tck, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co,
                  int lrelu, hipStream_t s) {
    if (Co != 16 || (Ci != 16 && Ci != 32) || ldx % 4 != 0 || ldy % 4 != 0) return MRDIS_EUNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W < 65536 || mrdis_opt(MRDIS_OPT_NOW16)) return MRDIS_EUNSUPPORTED;          // small maps: launch-bound either way
    const long long xb = 4LL * (((long long)N * H * W - 1) * ldx + Ci);
    if (xb >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    C16Params p{};
    p.x = x; p.w = w_tck; p.bias = bias; p.y = y; p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.ldx = ldx; p.ldy = ldy; p.lrelu = lrelu;
    p.tilesY = mrdis_cdiv(H, C16_TH); p.tilesX = mrdis_cdiv(W, C16_TW);
    const long long nt = (long long)N * p.tilesY * p.tilesX;
    if (nt > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.ntiles = (int)nt; p.x_bytes = (unsigned)xb;
    const int grid = p.ntiles < 512 ? p.ntiles : 512;
    const size_t lds = sizeof(float) * (size_t)C16_NPX * (Ci + 4);
    if (Ci == 32 && (mrdis_opt(MRDIS_OPT_SPLIT6) == 1 || mrdis_opt(MRDIS_OPT_SPLIT6) == 5)) {      // (5: this kernel only)
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)conv3x3_c16_split6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess) return MRDIS_ELAUNCH;
            attr_set = true;
        }
        mrdis_count(MRDIS_CNT_SPLIT6_C16);
        MRDIS_LAUNCH(conv3x3_c16_split6_kernel, dim3(grid), dim3(256), (size_t)C16_NPX * 208, s, p);
    } else
    if (Ci == 32) MRDIS_LAUNCH((conv3x3_c16_kernel<2>), dim3(grid), dim3(256), lds, s, p);
    else MRDIS_LAUNCH((conv3x3_c16_kernel<1>), dim3(grid), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
