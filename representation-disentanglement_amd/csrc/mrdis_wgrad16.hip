// mrdis_wgrad16.hip -- weight (+ bias) gradient of a 3x3 / stride 1 / pad 1 layer with Cout <= 16 (sp6.out 32 -> 16 at
// full resolution: 16 calls per step).  A 32x32 MFMA tile is half empty there (wgrad_dma_kernel: 50 TF/s); this is the
// 2-D form of wgrad3d16_kernel (mrdis_conv3d.hip): v_mfma_f32_16x16x4_f32 with A = 16 input channels x 4 positions,
// B = 4 positions x 16 couts; a wave keeps the accumulators of all 9 taps (36 registers), so one dy read feeds 9 MFMAs
// and every A operand is one ds_read of the halo'd x box.  The waves of a workgroup split the 128 positions (8 x 16) of a
// box; boxes are walked with a grid stride (split-K) with the next box's x / dy in flight in registers; Ci > 16 runs as
// 16-channel slices (one workgroup column each); in-block wave reduction, fixed-order slab reduction.
#include "mrdis_common.h"
#include <stdlib.h>

struct Wgrad16Params {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, H, W, Ci, ldx, Co, lddy;
    int tilesA, tilesB, numTiles, nCi, splits;
};

#define W16_TH 8
#define W16_TW 16
#define W16_RW (W16_TW + 2)
#define W16_NPX ((W16_TH + 2) * W16_RW)

template <int NS>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(const Wgrad16Params p) {      // (two waves per SIMD = 256 registers: with the default 512 hipcc parks the 36 accumulators in AGPRs between iterations, 72 moves per box)
    // NS = 16-channel slices of x handled by one workgroup (2 for Ci = 32: dy is staged and read once for both)
    constexpr int S = NS == 1 ? 16 : 48, QX = 4 * NS;           // pixel pitch = 16 mod 64: the four positions of a k-step land 16 banks apart
    constexpr int XR = (W16_NPX * QX + 255) / 256;          // 180 pixels x QX float4
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;                                  // [128][16]
    float* xs = smem + 2048;                            // [pixel][17]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int lb = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int ncol = p.nCi / NS;                       // workgroup columns
    const int cic = lb % ncol, split = lb / ncol;
    const int c_lo = cic * 16 * NS;

    int loff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) loff[t] = ((t / 3) * W16_RW + (t % 3)) * S + l16;
    bool ci_ok[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) ci_ok[sl] = (c_lo + 16 * sl + l16) < p.Ci;
    int tin[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int m = wave * 32 + g * 4 + kq;
        tin[g] = ((m / W16_TW) * W16_RW + (m % W16_TW)) * S;
    }
    int xl[XR], xc[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 256;
        xl[it] = -1; xc[it] = 0;
        if (idx < W16_NPX * QX) { const int pi = idx / QX, q = idx % QX; xl[it] = pi * S + 4 * q; xc[it] = ((pi / W16_RW) << 10) | (pi % W16_RW); }
    }
    const int qx = c_lo + (tid % QX) * 4;             // 256 % QX == 0: a thread's items share their channel quad
    const bool qx_ok = qx < p.Ci;
    int yc[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) { const int m = (tid + it * 256) >> 2; yc[it] = ((m / W16_TW) << 10) | (m % W16_TW); }
    const int qy = (tid & 3) * 4;
    const bool qy_ok = qy < p.Co;

    f32x4 acc[NS][9];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[sl][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    float4 xr[XR], yr[2];
    auto load_box = [&](int box) {
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA;
        const int n = tt / p.tilesA;
        const int a0 = ta * W16_TH, b0 = tb * W16_TW;
        const float* __restrict__ xn = p.x + (long long)n * p.H * p.W * p.ldx + qx;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int h = a0 - 1 + (xc[it] >> 10), w_ = b0 - 1 + (xc[it] & 1023);
            if (xl[it] >= 0 && qx_ok && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                xr[it] = *reinterpret_cast<const float4*>(xn + ((long long)h * p.W + w_) * p.ldx);
        }
        const float* __restrict__ dyn = p.dy + (long long)n * p.H * p.W * p.lddy + qy;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            yr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int a = a0 + (yc[it] >> 10), b = b0 + (yc[it] & 1023);
            if (qy_ok && a < p.H && b < p.W) yr[it] = *reinterpret_cast<const float4*>(dyn + ((long long)a * p.W + b) * p.lddy);
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it)
            if (xl[it] >= 0) *reinterpret_cast<float4*>(xs + xl[it]) = xr[it];
#pragma unroll
        for (int it = 0; it < 2; ++it) *reinterpret_cast<float4*>(dys + 4 * (tid + it * 256)) = yr[it];
    };

    int box = split;
    if (box < p.numTiles) load_box(box);
    store_box();
    __syncthreads();
    for (; box < p.numTiles; box += p.splits) {
        const int nxt = box + p.splits;
        if (nxt < p.numTiles) load_box(nxt);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float bv = dys[(wave * 32 + g * 4 + kq) * 16 + l16];
            bsum += bv;
            const float* xa = xs + tin[g];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                float av[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) av[t] = ci_ok[sl] ? xa[loff[t] + 16 * sl] : 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[sl][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc[sl][t], 0, 0, 0);
            }
        }
        if (nxt < p.numTiles) { __syncthreads(); store_box(); __syncthreads(); }
    }
    // cross-wave reduction through LDS (fixed order), then slab[split][cic][tap][16 ci][16 co]
    float* red = smem;                                  // [4 waves][9][256] = 9216 floats (host sizes the LDS for it)
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        float* out = p.slab + (((long long)split * p.nCi + cic * NS + sl) * 9) * 256;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 9 + t) * 256 + (4 * kq + r) * 16 + l16] = acc[sl][t][r];
        __syncthreads();
        for (int i = tid; i < 9 * 256; i += 256)
            out[i] = (red[i] + red[9 * 256 + i]) + (red[2 * 9 * 256 + i] + red[3 * 9 * 256 + i]);
    }
    if (p.bias_slab != nullptr && cic == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        __syncthreads();
        if (kq == 0) red[wave * 16 + l16] = bsum;
        __syncthreads();
        if (tid < 16) p.bias_slab[(long long)split * 16 + tid] = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
    }
}

// ---- six-product form (option split6; Ci = 32, Co = 16: sp6.out): the same weight gradient on the bf16 matrix pipe at fp32 accuracy.  wgrad16_kernel is
// bound by its fp32 MFMAs (19.3 GFLOP at 88 TF/s; the layer moves 402 MB: 75 us at the narrow layers' streaming rate).  Here the reduction axis -- 32
// consecutive positions of a box row -- is the K of v_mfma_f32_16x16x32_bf16, both operands are three bf16 terms (v = hi + mid + lo, split once per element
// on its way into LDS) and the six products of order <= 2 are summed in fp32.  The tap shift is put on the NARROW operand:
//     dW[t][ci][co] = sum_q x[q][ci] dy[q - off(t)][co],
// so the x operand of a row (2 ci tiles x 3 terms, transposing reads ds_read_b64_tr_b16 from [pixel][32 ch] planes) is read ONCE for the nine taps, and only
// the 32-byte dy rows (halo'd 10 x 34 box, [pixel][16 co] planes) are re-read per tap: 66 LDS reads per 108 MFMAs of a row.  A wave keeps all 9 x 2
// accumulators (72 VGPRs); the four waves of a workgroup take rows r, r + 4 of an 8 x 32 box; boxes are walked with a grid stride with the next box in flight
// in registers.  Slab layout, in-block wave reduction and the slab reduction launch are wgrad16_kernel's (a workgroup writes both 16-channel slices).
typedef __bf16 w16_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 w16_bf16x4 __attribute__((ext_vector_type(4)));
typedef short w16_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned w16_u32x4 __attribute__((ext_vector_type(4)));
struct Wgrad16SParams {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, H, W, ldx, lddy;
    int tilesA, tilesB, numTiles, splits;
    unsigned x_bytes, dy_bytes;
};
#define W6_TH 8
#define W6_TW 32
#define W6_YW (W6_TW + 2)
#define W6_YPX ((W6_TH + 2) * W6_YW)
__global__ __launch_bounds__(256, 2) void wgrad16_split6_kernel(const Wgrad16SParams p) {
    constexpr int XPX = W6_TH * W6_TW;                 // 256 x pixels (64-byte rows per term), 340 dy pixels (32-byte rows per term)
    constexpr int XPLANE = XPX * 64, YPLANE = W6_YPX * 32;
    constexpr int XR = 8, YR = 6;                      // 16-byte fp32 pieces per thread: 256 x 8 / 256, ceil(340 x 4 / 256)
    constexpr unsigned OOB = 0xfffffff0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    unsigned char* xs = smem6;                         // [3 terms][256 px][32 ch] bf16
    unsigned char* ys = smem6 + 3 * XPLANE;            // [3 terms][340 px][16 co] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int split = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);

    // transposing reads: lane (q4, p4) of a 16-lane group supplies the address of pixel-row q4, columns 4 p4 .. + 3 of the group's 4 x 16 block and
    // receives column l16 (a channel / a cout) of the four pixel rows; group kq covers positions 8 kq .. 8 kq + 7 of the k-step in two reads
    const int q4 = l16 >> 2, p4 = l16 & 3;
    const int xoff = (8 * kq + q4) * 64 + 8 * p4;      // + (row * 32) * 64 + 32 mt (channels 16 mt ..) + 4 * 64 (second read) + term * XPLANE
    const int yoff = (8 * kq + q4) * 32 + 8 * p4;      // + ((row - ty + 2) * 34 + 2 - tx) * 32 + 4 * 32 + term * YPLANE

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float bs4[4] = {0.f, 0.f, 0.f, 0.f};              // this thread's share of the bias gradient: couts 4 (tid & 3) .. + 3

    w16_u32x4 xr[XR], yr[YR];
    auto load_box = [&](int box) {                    // box >= numTiles: every offset out of range -> zeros
        const bool on = box < p.numTiles;
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA;
        const int n = tt / p.tilesA;
        const int a0 = ta * W6_TH, b0 = tb * W6_TW;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 3, q = idx & 7;
            const int h = a0 + (px >> 5), w_ = b0 + (px & 31);
            const bool ok = on && h < p.H && w_ < p.W;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldx + 4 * q) : OOB), 0, 0);
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 2, q = idx & 3;
            const int ly = px / W6_YW, lx = px - ly * W6_YW;
            const int h = a0 - 1 + ly, w_ = b0 - 1 + lx;
            const bool ok = on && px < W6_YPX && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            yr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)(ok ? 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.lddy + 4 * q) : OOB), 0, 0);
        }
    };
    auto split4 = [](const w16_u32x4& v, w16_bf16x4& hi, w16_bf16x4& mid, w16_bf16x4& lo) {
        const float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const __bf16 h_ = (__bf16)f[c]; const float r1 = f[c] - (float)h_; const __bf16 m_ = (__bf16)r1;
            hi[c] = h_; mid[c] = m_; lo[c] = (__bf16)(r1 - (float)m_);
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 3, q = idx & 7;
            w16_bf16x4 hi, mid, lo; split4(xr[it], hi, mid, lo);
            unsigned char* d = xs + px * 64 + 8 * q;
            *reinterpret_cast<w16_bf16x4*>(d) = hi; *reinterpret_cast<w16_bf16x4*>(d + XPLANE) = mid; *reinterpret_cast<w16_bf16x4*>(d + 2 * XPLANE) = lo;
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 2, q = idx & 3;
            if (px >= W6_YPX) continue;
            const int ly = px / W6_YW, lx = px - ly * W6_YW;
            if (ly >= 1 && ly <= W6_TH && lx >= 1 && lx <= W6_TW) {      // the box's own positions: the bias gradient (fp32, before the split)
                bs4[0] += __uint_as_float(yr[it].x); bs4[1] += __uint_as_float(yr[it].y); bs4[2] += __uint_as_float(yr[it].z); bs4[3] += __uint_as_float(yr[it].w);
            }
            w16_bf16x4 hi, mid, lo; split4(yr[it], hi, mid, lo);
            unsigned char* d = ys + px * 32 + 8 * q;
            *reinterpret_cast<w16_bf16x4*>(d) = hi; *reinterpret_cast<w16_bf16x4*>(d + YPLANE) = mid; *reinterpret_cast<w16_bf16x4*>(d + 2 * YPLANE) = lo;
        }
    };
    int box = split;
    load_box(box);
    store_box();
    __syncthreads();
    for (; box < p.numTiles; box += p.splits) {
        load_box(box + p.splits);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = wave + 4 * rr;
            // x operand of the row: [ci tile mt][term], read once for the nine taps
            w16_bf16x8 ax[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    const unsigned char* a = xs + term * XPLANE + row * (32 * 64) + xoff + 32 * mt;
                    union { w16_bf16x8 v; w16_s16x4 h[2]; } u;
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(a));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(a + 4 * 64));
                    ax[mt][term] = u.v;
                }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t % 3;
                w16_bf16x8 by[3];
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    const unsigned char* b = ys + term * YPLANE + ((row - ty + 2) * W6_YW + 2 - tx) * 32 + yoff;
                    union { w16_bf16x8 v; w16_s16x4 h[2]; } u;
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(b));
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(b + 4 * 32));
                    by[term] = u.v;
                }
                // six products of order <= 2 (terms 0 = hi, 1 = mid, 2 = lo), smallest first
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][1], by[1], acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][2], by[0], acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][0], by[2], acc[t][mt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][1], by[0], acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][0], by[1], acc[t][mt], 0, 0, 0);
                    acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][0], by[0], acc[t][mt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        store_box();
        __syncthreads();
    }
    // cross-wave reduction through LDS (fixed order), then slab[split][ci tile][tap][16 ci][16 co] -- as wgrad16_kernel
    float* red = reinterpret_cast<float*>(smem6);      // [4 waves][9][256]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float* out = p.slab + (((long long)split * 2 + mt) * 9) * 256;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 9 + t) * 256 + (4 * kq + r) * 16 + l16] = acc[t][mt][r];
        __syncthreads();
        for (int i = tid; i < 9 * 256; i += 256)
            out[i] = (red[i] + red[9 * 256 + i]) + (red[2 * 9 * 256 + i] + red[3 * 9 * 256 + i]);
    }
    if (p.bias_slab != nullptr) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[c * 256 + tid] = bs4[c];      // thread tid holds couts 4 (tid & 3) + c
        __syncthreads();
        if (tid < 16) {
            const int qd = tid >> 2, c = tid & 3;
            float t_ = 0.f;
            for (int k = qd; k < 256; k += 4) t_ += red[c * 256 + k];
            p.bias_slab[(long long)split * 16 + tid] = t_;
        }
    }
}

// ---- the same weight gradient on bf16 views (MRDIS_DT_BF16: x (N, H, W, 32) and dy (N, H, W, 16) stored in bf16): the structure of wgrad16_split6_kernel
// with ONE term per operand -- the 16-byte pieces of both tensors go from global memory into the [pixel][C] LDS images as they are (no split, no conversion),
// 18 bf16 MFMAs per 32-position row.  The generic bf16 kernel (bwgrad3_kernel<1, 1>) keeps a 32 x 32 accumulator block per tap for this 32 x 16 problem and
// runs it in 81 us; the layer moves 201 MB.
__global__ __launch_bounds__(256, 2) void wgrad16_bf16_kernel(const Wgrad16SParams p) {
    constexpr int XPX = W6_TH * W6_TW;
    constexpr int XR = 4, YR = 3;                      // 16-byte bf16 pieces per thread: 256 x 4 / 256, ceil(340 x 2 / 256)
    constexpr unsigned OOB = 0xfffffff0u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
    unsigned char* xs = smem7;                         // [256 px][32 ch] bf16
    unsigned char* ys = smem7 + XPX * 64;              // [340 px][16 co] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int split = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const int q4 = l16 >> 2, p4 = l16 & 3;
    const int xoff = (8 * kq + q4) * 64 + 8 * p4;
    const int yoff = (8 * kq + q4) * 32 + 8 * p4;

    f32x4 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float bs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // this thread's share of the bias gradient: couts 8 (tid & 1) .. + 7

    w16_u32x4 xr[XR], yr[YR];
    auto load_box = [&](int box) {
        const bool on = box < p.numTiles;
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA;
        const int n = tt / p.tilesA;
        const int a0 = ta * W6_TH, b0 = tb * W6_TW;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 2, q = idx & 3;
            const int h = a0 + (px >> 5), w_ = b0 + (px & 31);
            const bool ok = on && h < p.H && w_ < p.W;
            xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? 2u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldx + 8 * q) : OOB), 0, 0);
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 1, q = idx & 1;
            const int ly = px / W6_YW, lx = px - ly * W6_YW;
            const int h = a0 - 1 + ly, w_ = b0 - 1 + lx;
            const bool ok = on && px < W6_YPX && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            yr[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)(ok ? 2u * (unsigned)(((n * p.H + h) * p.W + w_) * p.lddy + 8 * q) : OOB), 0, 0);
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 2, q = idx & 3;
            *reinterpret_cast<w16_u32x4*>(xs + px * 64 + 16 * q) = xr[it];
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 1, q = idx & 1;
            if (px >= W6_YPX) continue;
            const int ly = px / W6_YW, lx = px - ly * W6_YW;
            if (ly >= 1 && ly <= W6_TH && lx >= 1 && lx <= W6_TW) {      // the box's own positions: the bias gradient (fp32 sum of the bf16 values)
                const unsigned u[4] = {yr[it].x, yr[it].y, yr[it].z, yr[it].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) { bs8[2 * k] += __uint_as_float(u[k] << 16); bs8[2 * k + 1] += __uint_as_float(u[k] & 0xffff0000u); }
            }
            *reinterpret_cast<w16_u32x4*>(ys + px * 32 + 16 * q) = yr[it];
        }
    };

    int box = split;
    load_box(box);
    store_box();
    __syncthreads();
    for (; box < p.numTiles; box += p.splits) {
        load_box(box + p.splits);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = wave + 4 * rr;
            w16_bf16x8 ax[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned char* a = xs + row * (32 * 64) + xoff + 32 * mt;
                union { w16_bf16x8 v; w16_s16x4 h[2]; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(a));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(a + 4 * 64));
                ax[mt] = u.v;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t % 3;
                const unsigned char* b = ys + ((row - ty + 2) * W6_YW + 2 - tx) * 32 + yoff;
                union { w16_bf16x8 v; w16_s16x4 h[2]; } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(b));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w16_s16x4 __attribute__((address_space(3)))*)(b + 4 * 32));
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], u.v, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[1], u.v, acc[t][1], 0, 0, 0);
            }
        }
        __syncthreads();
        store_box();
        __syncthreads();
    }
    float* red = reinterpret_cast<float*>(smem7);      // [4 waves][9][256] (host: the LDS is at least that large)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float* out = p.slab + (((long long)split * 2 + mt) * 9) * 256;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 9 + t) * 256 + (4 * kq + r) * 16 + l16] = acc[t][mt][r];
        __syncthreads();
        for (int i = tid; i < 9 * 256; i += 256)
            out[i] = (red[i] + red[9 * 256 + i]) + (red[2 * 9 * 256 + i] + red[3 * 9 * 256 + i]);
    }
    if (p.bias_slab != nullptr) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) red[c * 256 + tid] = bs8[c];      // thread tid holds couts 8 (tid & 1) + c
        __syncthreads();
        if (tid < 16) {
            const int hf = tid >> 3, c = tid & 7;
            float t_ = 0.f;
            for (int k = hf; k < 256; k += 2) t_ += red[c * 256 + k];
            p.bias_slab[(long long)split * 16 + tid] = t_;
        }
    }
}

__global__ void wgrad16_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Ci, int Co, int nCi, int nslab,
                                      const float* __restrict__ bslab, float* __restrict__ dbias, int accumulate_bias) {
    __shared__ float red[16][65];
    const int total = 9 * Ci * Co;
    const int nout = total + (dbias != nullptr ? Co : 0);
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int SL = blockDim.y, y = threadIdx.y;
    float s_ = 0.f;
    if (i < total) {
        const int co = i % Co;
        const int r = i / Co;
        const int ci = r % Ci, t = r / Ci;
        const long long stride = (long long)nCi * 9 * 256;
        const float* src = slab + ((long long)(ci >> 4) * 9 + t) * 256 + (ci & 15) * 16 + co + (long long)y * stride;
        for (int k = y; k < nslab; k += SL, src += (long long)SL * stride) s_ += *src;
    } else if (i < nout) {
        const int co = i - total;
        for (int k = y; k < nslab; k += SL) s_ += bslab[(long long)k * 16 + co];
    }
    red[y][threadIdx.x] = s_;
    __syncthreads();
    if (y == 0 && i < nout) {
        float t = 0.f;
        for (int k = 0; k < SL; ++k) t += red[k][threadIdx.x];
        if (i < total) dw[i] = t;
        else { const int co = i - total; dbias[co] = accumulate_bias ? dbias[co] + t : t; }
    }
}

static bool plan_wgrad16(Wgrad16Params& p, int N, int H, int W, int Ci, int Co) {
    if (Co > 16 || Co <= 4 || Co % 4 != 0 || Ci % 16 != 0) return false;               // Cout <= 4: wgrad_thin_dma_kernel
    if ((long long)N * H * W < 100000 || mrdis_opt(MRDIS_OPT_NOW16)) return false;       // small maps: the generic kernel's slabs are cheaper
    p = Wgrad16Params{};
    p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co;
    p.tilesA = mrdis_cdiv(H, W16_TH); p.tilesB = mrdis_cdiv(W, W16_TW);
    const long long nt = (long long)N * p.tilesA * p.tilesB;
    if (nt > 0x7fffffffLL) return false;
    p.numTiles = (int)nt;
    p.nCi = Ci / 16;
    int splits = 1024 / p.nCi;
    if (splits > p.numTiles) splits = p.numTiles;
    if (splits < 1) splits = 1;
    p.splits = splits;
    return true;
}

size_t mrdis_wgrad16_workspace(int N, int H, int W, int Ci, int Co) {
    Wgrad16Params p;
    if (!plan_wgrad16(p, N, H, W, Ci, Co)) return 0;
    return sizeof(float) * ((size_t)p.splits * p.nCi * 9 * 256 + (size_t)p.splits * 16) + 256;
}

// returns MRDIS_EUNSUPPORTED when the layer is outside what this kernel covers
int mrdis_run_wgrad16(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                      size_t workspace_bytes, int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s) {
    Wgrad16Params p;
    if (!plan_wgrad16(p, N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    if (ldx % 4 != 0 || lddy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_wgrad16_workspace(N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy;
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + (size_t)p.splits * p.nCi * 9 * 256 : nullptr;
    if (Ci == 32 && Co == 16 && (mrdis_opt(MRDIS_OPT_SPLIT6) == 1 || mrdis_opt(MRDIS_OPT_SPLIT6) == 6)) {      // six bf16 products per fp32 product (6: this kernel only)
        const long long xb = 4LL * (((long long)N * H * W - 1) * ldx + Ci), yb = 4LL * (((long long)N * H * W - 1) * lddy + Co);
        if (xb < 0x7fffffffLL && yb < 0x7fffffffLL) {
            Wgrad16SParams q{};
            q.x = x; q.dy = dy; q.slab = p.slab; q.bias_slab = p.bias_slab; q.N = N; q.H = H; q.W = W; q.ldx = ldx; q.lddy = lddy;
            q.tilesA = mrdis_cdiv(H, W6_TH); q.tilesB = mrdis_cdiv(W, W6_TW);
            const long long nt6 = (long long)N * q.tilesA * q.tilesB;
            q.numTiles = (int)nt6; q.splits = p.splits < q.numTiles ? p.splits : q.numTiles;      // (the slabs of plan_wgrad16: [splits][2][9][256] + [splits][16])
            q.x_bytes = (unsigned)xb; q.dy_bytes = (unsigned)yb;
            const size_t lds6 = 3 * (size_t)(W6_TH * W6_TW * 64) + 3 * (size_t)(W6_YPX * 32);
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute((const void*)wgrad16_split6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return MRDIS_ELAUNCH;
                attr_set = true;
            }
            mrdis_count(MRDIS_CNT_SPLIT6_WGRAD16);
            MRDIS_LAUNCH(wgrad16_split6_kernel, dim3(q.splits), dim3(256), lds6, s, q);
            MRDIS_CHECK_LAUNCH();
            const long long nout6 = 9LL * Ci * Co + (dbias ? Co : 0);
            int SL6 = 1;
            while (SL6 < 16 && SL6 * 8 <= q.splits) SL6 <<= 1;
            MRDIS_LAUNCH(wgrad16_reduce_kernel, dim3(mrdis_cdiv(nout6, 64)), dim3(64, SL6), 0, s, p.slab, dw_tck, Ci, Co, 2, q.splits, p.bias_slab, dbias, accumulate_bias);
            MRDIS_CHECK_LAUNCH();
            return MRDIS_OK;
        }
    }
    const size_t lds = sizeof(float) * (size_t)(4 * 9 * 256);           // reduction buffer (36 KB) >= dys + x box (19.5 KB)
    // (two slices per workgroup -- dy staged once, wgrad16_kernel<2> -- measured slower: 338 vs 289 us on 32 -> 16 at 256x256, B = 32;
    //  the kernel lives on workgroup-level overlap of its staging and MFMA phases, and half as many workgroups overlap less)
    MRDIS_LAUNCH(wgrad16_kernel<1>, dim3(p.splits * p.nCi), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    const long long nout = 9LL * Ci * Co + (dbias ? Co : 0);
    int SL = 1;
    while (SL < 16 && SL * 8 <= p.splits) SL <<= 1;
    MRDIS_LAUNCH(wgrad16_reduce_kernel, dim3(mrdis_cdiv(nout, 64)), dim3(64, SL), 0, s, p.slab, dw_tck, Ci, Co, p.nCi, p.splits,
                       p.bias_slab, dbias, accumulate_bias);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// bf16 views (MRDIS_DT_BF16), Ci = 32, Co = 16, 3x3 s1 p1, large maps: wgrad16_bf16_kernel; MRDIS_EUNSUPPORTED elsewhere (the caller runs the generic bf16 kernel)
int mrdis_run_wgrad16_bf16(const void* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s) {
    Wgrad16Params p;
    if (Ci != 32 || Co != 16 || !plan_wgrad16(p, N, H, W, Ci, Co) || mrdis_opt(MRDIS_OPT_MODE) == 3030) return MRDIS_EUNSUPPORTED;      // (debug_mode 3030: the generic kernel, for A/B)
    if (ldx % 8 != 0 || lddy % 8 != 0 || ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_wgrad16_workspace(N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    const long long xb = 2LL * (((long long)N * H * W - 1) * ldx + Ci), yb = 2LL * (((long long)N * H * W - 1) * lddy + Co);
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    Wgrad16SParams q{};
    q.x = reinterpret_cast<const float*>(x); q.dy = reinterpret_cast<const float*>(dy);
    q.slab = reinterpret_cast<float*>(workspace); q.bias_slab = dbias ? q.slab + (size_t)p.splits * p.nCi * 9 * 256 : nullptr;
    q.N = N; q.H = H; q.W = W; q.ldx = ldx; q.lddy = lddy;
    q.tilesA = mrdis_cdiv(H, W6_TH); q.tilesB = mrdis_cdiv(W, W6_TW);
    const long long nt = (long long)N * q.tilesA * q.tilesB;
    q.numTiles = (int)nt; q.splits = p.splits < q.numTiles ? p.splits : q.numTiles;
    q.x_bytes = (unsigned)xb; q.dy_bytes = (unsigned)yb;
    size_t lds = (size_t)(W6_TH * W6_TW * 64) + (size_t)(W6_YPX * 32);
    const size_t red = sizeof(float) * (size_t)(4 * 9 * 256);
    if (lds < red) lds = red;
    MRDIS_LAUNCH(wgrad16_bf16_kernel, dim3(q.splits), dim3(256), lds, s, q);
    MRDIS_CHECK_LAUNCH();
    const long long nout = 9LL * Ci * Co + (dbias ? Co : 0);
    int SL = 1;
    while (SL < 16 && SL * 8 <= q.splits) SL <<= 1;
    MRDIS_LAUNCH(wgrad16_reduce_kernel, dim3(mrdis_cdiv(nout, 64)), dim3(64, SL), 0, s, q.slab, dw_tck, Ci, Co, 2, q.splits, q.bias_slab, dbias, accumulate_bias);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
