// mrdis_wgrad_s2.hip -- weight (+ bias) gradient of the stride-2 first layers: Cin <= 7 image channels, 3x3 or 4x4 taps, pad 1,
// Cout = 16 / 32 (ana_enc.down_1 7 -> 32 k4 and mod_enc.conv1 7 -> 16 k3 at full resolution, model.py:2150 / :2374; the input is a
// 7-channel slice of the 28-channel batch tensor, so ldx = 28 and the base is not 16-byte aligned).
//
// The layer is HBM-bound (59 MB of x + 34-67 MB of dy per call, 1-4 GFLOP); the generic split-K kernel (wgrad_kernel<1>) spends
// 190 us per call on scalar staging.  Here the GEMM is dW[m = (tap, ci)][co] = sum_pixels xpatch[m][pixel] * dy[pixel][co]:
//   v_mfma_f32_16x16x4_f32, A = 16 (tap, ci) rows x 4 output pixels, B = 4 pixels x 16 couts, all (tap, ci) tiles of a wave live in
//   registers (k4: 7 x 2 tiles = 56 accumulators), so one dy read feeds 7 MFMAs and one x read 2.
//   A workgroup walks output rows (n, oy) with a grid stride; the k input rows of an output row sit in LDS as the memory image
//   [row][pixel + 1][ci] (one zero pixel each side), so A(m, pixel) is ONE ds_read at  row(m) * pitch + (tx(m) * Ci + ci(m)) + 2 Ci * ox:
//   consecutive m are consecutive words, the pitch continues that sequence across tap rows (pitch = k * Ci mod 64), the four pixel
//   groups of a wave sit 2 Ci = 14 banks apart: conflict-free up to the two rows shared by neighbouring groups.
//   dy rows are staged [pixel][48 | 16] (48: the four pixel groups land 16 banks apart).
//   The next output row's x / dy are in flight in registers while the MFMAs of the current one run.
// In-block wave reduction through LDS, fixed-order slab reduction: deterministic.
#include "mrdis_common.h"

struct WgradS2Params {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, H, W, Ci, ldx, Co, lddy, Hout, Wout;
    int rowp;                 // LDS pitch of an x row, floats
    int segs, R, splits;      // workgroups per image, output rows per workgroup, workgroups in total
    int M;                    // taps * Ci rows of the gradient
    unsigned x_bytes, dy_bytes;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
namespace {
constexpr unsigned S2_OOB = 0xfffffff0u;
constexpr int S2_XJ = 7;      // x items per thread and input row: W * Ci <= 1792
template <int V_> struct IC { static constexpr int value = V_; };
}

template <int KS, int NT>
__global__ __launch_bounds__(256) void wgrad_s2_kernel(const WgradS2Params p) {
    constexpr int MT = (KS * KS * 7 + 15) / 16;       // 7 (k4) / 4 (k3) tiles of 16 (tap, ci) rows
    constexpr int DYP = NT == 2 ? 48 : 16;
    constexpr int YJ = NT == 2 ? 4 : 2;               // dy float4 items per thread and row: Wout * Co / 4 <= 256 * YJ
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                 // [4][rowp]: two pairs of input rows
    float* dys = smem + 4 * p.rowp;                   // [Wout][DYP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int split = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int Ci = p.Ci, Co = NT * 16;
    const int n = split / p.segs, oy0 = (split - n * p.segs) * p.R, oy1 = min(oy0 + p.R, p.Hout);      // a workgroup = R consecutive output rows of one image

    // A-operand offsets: row m = mt * 16 + l16 = (ty * KS + tx) * Ci + ci
    // (LDS row of tap row ty at an even output row: ty; at an odd one the two pairs have swapped places: ty ^ 2)
    int aoff[MT], aflip[MT]; bool aok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l16;
        aok[mt] = m < p.M;
        const int tap = aok[mt] ? m / Ci : 0, ci = aok[mt] ? m - tap * Ci : 0;
        const int ty = tap / KS;
        aoff[mt] = ty * p.rowp + (tap % KS) * Ci + ci + 2 * Ci * kq;
        aflip[mt] = ((ty ^ 2) - ty) * p.rowp;
    }
    const int boff = kq * DYP + l16;

    // staging roles
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    unsigned xg[S2_XJ]; int xl[S2_XJ];
#pragma unroll
    for (int j = 0; j < S2_XJ; ++j) {
        const int idx = tid + 256 * j;
        xg[j] = S2_OOB; xl[j] = -1;
        if (idx < p.W * Ci) { const int pix = idx / Ci, c = idx - pix * Ci; xg[j] = 4u * (unsigned)(pix * p.ldx + c); xl[j] = (pix + 1) * Ci + c; }
    }
    const int cq = Co >> 2;
    unsigned yg[YJ]; int yl[YJ];
#pragma unroll
    for (int j = 0; j < YJ; ++j) {
        const int idx = tid + 256 * j;
        yg[j] = S2_OOB; yl[j] = -1;
        if (idx < p.Wout * cq) { const int pix = idx / cq, q = idx - pix * cq; yg[j] = 4u * (unsigned)(pix * p.lddy + 4 * q); yl[j] = pix * DYP + 4 * q; }
    }
    // the zero pixels either side of every x row (and the pitch padding) are written once
    for (int i = tid; i < 4 * p.rowp; i += 256) xs[i] = 0.f;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

    // input rows travel in pairs: pair j = rows (2 j - 1, 2 j) lives in LDS rows 2 (j & 1) + (0 | 1); output row oy reads pairs oy and oy + 1
    // (rows 2 oy - 1 .. 2 oy + 2; 3x3 taps leave the last one unread), so stepping to oy + 1 replaces pair oy by pair oy + 2:
    // two new rows per output row instead of KS.  Rows outside the image come back as zeros (offset out of range).
    // Two register sets: what output row oy + 1 needs (pair oy + 2, dy row oy + 1) and what oy + 2 needs are both in flight while oy is
    // computed -- one output row of MFMAs is shorter than the memory latency.
    float xr[2][2][S2_XJ];
    u32x4 yr[2][YJ];
    auto load_set = [&](auto S_, int j, int oy) {     // pair j and dy row oy; nothing to fetch -> offsets out of range -> zeros, no branch
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int iy = 2 * j - 1 + h;
            const bool ok = j <= oy1 && (unsigned)iy < (unsigned)p.H;
            const unsigned base = 4u * (unsigned)((n * p.H + iy) * p.W * p.ldx);           // host: the tensor is < 2^31 bytes
#pragma unroll
            for (int jj = 0; jj < S2_XJ; ++jj)
                xr[S][h][jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)((ok && xl[jj] >= 0) ? base + xg[jj] : S2_OOB), 0, 0));
        }
        const bool on = oy >= 0 && oy < oy1;
        const unsigned ybase = 4u * (unsigned)((n * p.Hout + oy) * p.Wout * p.lddy);
#pragma unroll
        for (int j2 = 0; j2 < YJ; ++j2) yr[S][j2] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)((on && yl[j2] >= 0) ? ybase + yg[j2] : S2_OOB), 0, 0);
    };
    auto store_set = [&](auto S_, int j, bool with_dy) {
        constexpr int S = decltype(S_)::value;
        float* d = xs + 2 * (j & 1) * p.rowp;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < S2_XJ; ++jj)
                if (xl[jj] >= 0) d[h * p.rowp + xl[jj]] = xr[S][h][jj];
        if (with_dy) {
#pragma unroll
            for (int j2 = 0; j2 < YJ; ++j2)
                if (yl[j2] >= 0) *reinterpret_cast<u32x4*>(dys + yl[j2]) = yr[S][j2];
        }
    };
    const int nsteps = p.Wout >> 4;                   // host: Wout % 16 == 0; a wave takes every fourth group of 4 pixels
    auto compute = [&](int oy) {
        int ao[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ao[mt] = aoff[mt] + ((oy & 1) ? aflip[mt] : 0);
        // operands of pixel group i + 1 are read while the MFMAs of group i run (the last group re-reads itself: no branch)
        float bvn[NT], avn[MT];
        auto read_ops = [&](int i) {
            const int ox0 = 4 * (wave + 4 * i);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bvn[nt] = dys[ox0 * DYP + boff + 16 * nt];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) avn[mt] = xs[ao[mt] + 2 * Ci * ox0];
        };
        read_ops(0);
        for (int i = 0; i < nsteps; ++i) {
            float bv[NT], av[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bv[nt] = bvn[nt]; bsum[nt] += bv[nt]; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = aok[mt] ? avn[mt] : 0.f;
            read_ops(i + 1 < nsteps ? i + 1 : i);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
    };
    auto step = [&](auto S_, int oy) {                // set S holds (pair oy + 2, dy row oy + 1)
        compute(oy);
        __syncthreads();
        store_set(S_, oy + 2, true);
        load_set(S_, oy + 4, oy + 3);
        __syncthreads();
    };

    load_set(IC<0>{}, oy0, -1);
    load_set(IC<1>{}, oy0 + 1, oy0);
    __syncthreads();                                  // the zero fill
    store_set(IC<0>{}, oy0, false);
    store_set(IC<1>{}, oy0 + 1, true);
    load_set(IC<0>{}, oy0 + 2, oy0 + 1);
    load_set(IC<1>{}, oy0 + 3, oy0 + 2);
    __syncthreads();
    for (int oy = oy0; oy < oy1; oy += 2) {           // an odd row count runs one more row on an all-zero dy row
        step(IC<0>{}, oy);
        step(IC<1>{}, oy + 1);
    }

    // cross-wave reduction, one (tap, ci) tile at a time: red[wave][nt][16 rows][16 couts]
    float* red = smem;
    float* out = p.slab + (long long)split * p.M * Co;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NT + nt) * 256 + (4 * kq + r) * 16 + l16] = acc[mt][nt][r];
        __syncthreads();
        for (int e = tid; e < NT * 256; e += 256) {
            const int nt = e >> 8, rr = (e >> 4) & 15, cc = e & 15, m = mt * 16 + rr;
            const float v = (red[e] + red[NT * 256 + e]) + (red[2 * NT * 256 + e] + red[3 * NT * 256 + e]);
            if (m < p.M) out[m * Co + nt * 16 + cc] = v;
        }
        __syncthreads();
    }
    if (p.bias_slab != nullptr) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float b = bsum[nt];
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (kq == 0) red[(wave * NT + nt) * 16 + l16] = b;
        }
        __syncthreads();
        if (tid < Co) {
            const int nt = tid >> 4, cc = tid & 15;
            p.bias_slab[(long long)split * Co + tid] = (red[nt * 16 + cc] + red[(NT + nt) * 16 + cc]) + (red[(2 * NT + nt) * 16 + cc] + red[(3 * NT + nt) * 16 + cc]);
        }
    }
}

// pad (rin, cin, rout, cout; all 0 = none): dw is [T][rout][cout] around the slabs' [T][rin][cin] -- the zero rows / columns of a filter stored in the
// 16-row or 16-column layout of the mixing launch are written here instead of by a pad of the result (total then counts the PADDED elements)
struct SlabPad { int rin, cin, rout, cout; };
__global__ __launch_bounds__(1024) void wgrad_s2_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int total, int Co, int nslab,
                                                               const float* __restrict__ bslab, float* __restrict__ dbias, int accumulate_bias, SlabPad pad) {
    // 32 outputs x 32 slab lanes per block; a lane walks its slabs four at a time (independent loads in flight), fixed order throughout
    __shared__ float red[32][33];
    const int nout = total + (dbias != nullptr ? Co : 0);
    const int i = blockIdx.x * 32 + threadIdx.x, y = threadIdx.y;
    const float* src = nullptr; long long stride = 0;
    if (i < total) {
        if (pad.rout == 0) { src = slab + i; stride = total; }
        else {
            const int c = i % pad.cout, tr = i / pad.cout, r = tr % pad.rout, t = tr / pad.rout;
            if (r < pad.rin && c < pad.cin) { src = slab + (t * pad.rin + r) * pad.cin + c; stride = (long long)(total / (pad.rout * pad.cout)) * pad.rin * pad.cin; }
        }
    }
    else if (i < nout) { src = bslab + (i - total); stride = Co; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (src != nullptr) {
        int k = y;
        for (; k + 96 < nslab; k += 128) {
            s0 += src[(long long)k * stride]; s1 += src[(long long)(k + 32) * stride];
            s2 += src[(long long)(k + 64) * stride]; s3 += src[(long long)(k + 96) * stride];
        }
        for (; k < nslab; k += 32) s0 += src[(long long)k * stride];
    }
    red[y][threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (y == 0 && i < nout) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
        if (i < total) dw[i] = t;
        else { const int co = i - total; dbias[co] = accumulate_bias ? dbias[co] + t : t; }
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// The same layers forward: y[pixel][co] = bias[co] + sum_m xpatch[m][pixel] w[m][co], m = (tap, ci).  Same staging (rolling input-row
// pairs, two register sets in flight); the filter is the A operand and lives in registers (k4, 32 couts: 28 k-steps x 2 = 56 VGPRs),
// the pixel operand B[k = m][pixel] is one ds_read per (k-step, 16-pixel tile) at  row(m) * pitch + (2 ox + tx(m)) * Ci + ci(m).
// D[cout][pixel]: lane = pixel, registers = 4 consecutive couts -> 16-byte stores.
struct ConvS2Params {
    const float* x; const float* w; const float* bias; float* y;
    int N, H, W, Ci, ldx, Co, ldy, Hout, Wout, lrelu;
    int rowp, segs, R, M;
    unsigned x_bytes;
};

template <int KS, int NT>
__global__ __launch_bounds__(256) void conv_s2_fwd_kernel(const ConvS2Params p) {
    constexpr int KSTEPS = (KS * KS * 7 + 3) / 4;     // 28 (k4) / 16 (k3) k-steps of 4 (tap, ci) rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                 // [4][rowp]: two pairs of input rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int split = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int Ci = p.Ci;
    const int n = split / p.segs, oy0 = (split - n * p.segs) * p.R, oy1 = min(oy0 + p.R, p.Hout);

    // A = filter: a[s][nt] = w[m = 4 s + kq][16 nt + l16]; B offsets of row m at an even output row, and which rows swap at an odd one
    float a[KSTEPS][NT];
    int moff[KSTEPS];
    unsigned hi = 0;                                  // bit s: tap row >= 2 (its LDS row moves down by two at odd output rows, the others up)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        const int m = 4 * s + kq;
        const bool ok = m < p.M;
        const int tap = ok ? m / Ci : 0, ci = ok ? m - tap * Ci : 0, ty = tap / KS;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) a[s][nt] = ok ? p.w[m * p.Co + 16 * nt + l16] : 0.f;
        moff[s] = ty * p.rowp + (tap % KS) * Ci + ci + 2 * Ci * l16;
        if (ty >= 2) hi |= 1u << s;
    }
    float4 bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        bv[nt] = p.bias ? make_float4(p.bias[16 * nt + 4 * kq], p.bias[16 * nt + 4 * kq + 1], p.bias[16 * nt + 4 * kq + 2], p.bias[16 * nt + 4 * kq + 3])
                        : make_float4(0.f, 0.f, 0.f, 0.f);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    unsigned xg[S2_XJ]; int xl[S2_XJ];
#pragma unroll
    for (int j = 0; j < S2_XJ; ++j) {
        const int idx = tid + 256 * j;
        xg[j] = S2_OOB; xl[j] = -1;
        if (idx < p.W * Ci) { const int pix = idx / Ci, c = idx - pix * Ci; xg[j] = 4u * (unsigned)(pix * p.ldx + c); xl[j] = (pix + 1) * Ci + c; }
    }
    for (int i = tid; i < 4 * p.rowp; i += 256) xs[i] = 0.f;

    float xr[2][2][S2_XJ];
    auto load_set = [&](auto S_, int j) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int iy = 2 * j - 1 + h;
            const bool ok = j <= oy1 && (unsigned)iy < (unsigned)p.H;
            const unsigned base = 4u * (unsigned)((n * p.H + iy) * p.W * p.ldx);
#pragma unroll
            for (int jj = 0; jj < S2_XJ; ++jj)
                xr[S][h][jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)((ok && xl[jj] >= 0) ? base + xg[jj] : S2_OOB), 0, 0));
        }
    };
    auto store_set = [&](auto S_, int j) {
        constexpr int S = decltype(S_)::value;
        float* d = xs + 2 * (j & 1) * p.rowp;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < S2_XJ; ++jj)
                if (xl[jj] >= 0) d[h * p.rowp + xl[jj]] = xr[S][h][jj];
    };
    const int ntile = p.Wout >> 4;                    // host: Wout % 16 == 0; wave w takes tiles w, w + 4, ...
    auto compute = [&](int oy) {
        if (oy >= oy1) return;                        // (wave-uniform; an odd row count runs the staging of one more row)
        const int flip = (oy & 1) ? 2 * p.rowp : 0;
        for (int t = wave; t < ntile; t += 4) {
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xt = xs + 32 * Ci * t;
            float bn = xt[moff[0] + ((hi & 1u) ? -flip : flip)];
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
                const float b = bn;
                if (s + 1 < KSTEPS) bn = xt[moff[s + 1] + (((hi >> (s + 1)) & 1u) ? -flip : flip)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][nt], b, acc[nt], 0, 0, 0);
            }
            const int ox = 16 * t + l16;
            float* dst = p.y + ((long long)(n * p.Hout + oy) * p.Wout + ox) * p.ldy + 4 * kq;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float4 o = make_float4(acc[nt][0] + bv[nt].x, acc[nt][1] + bv[nt].y, acc[nt][2] + bv[nt].z, acc[nt][3] + bv[nt].w);
                if (p.lrelu) { o.x = o.x > 0.f ? o.x : 0.2f * o.x; o.y = o.y > 0.f ? o.y : 0.2f * o.y; o.z = o.z > 0.f ? o.z : 0.2f * o.z; o.w = o.w > 0.f ? o.w : 0.2f * o.w; }
                *reinterpret_cast<float4*>(dst + 16 * nt) = o;
            }
        }
    };
    auto step = [&](auto S_, int oy) {                // set S holds pair oy + 2
        compute(oy);
        __syncthreads();
        store_set(S_, oy + 2);
        load_set(S_, oy + 4);
        __syncthreads();
    };
    load_set(IC<0>{}, oy0);
    load_set(IC<1>{}, oy0 + 1);
    __syncthreads();                                  // the zero fill
    store_set(IC<0>{}, oy0);
    store_set(IC<1>{}, oy0 + 1);
    load_set(IC<0>{}, oy0 + 2);
    load_set(IC<1>{}, oy0 + 3);
    __syncthreads();
    for (int oy = oy0; oy < oy1; oy += 2) {
        step(IC<0>{}, oy);
        step(IC<1>{}, oy + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------- 4 -> C, stride 1
// Weight (+ bias) gradient of the SPADE `si_layers` (4 -> 32 / 64 / 128, 3x3 s1 p1, model.py:2436; 48 calls per step on the 256^2 ..
// 64^2 maps): the same GEMM  dW[m = (tap, ci)][co] = sum_pixels xpatch[m][pixel] dy[pixel][co]  with 36 rows (three 16-row tiles) and
// NT = C / 16 cout tiles.  An output row needs input rows oy - 1 .. oy + 1: a four-slot ring in LDS (row iy lives in slot (iy + 1) & 3,
// memory image [pixel + 1][4 ch], one new row per output row), dy rows [pixel][C + 16]; two register sets in flight.
struct WgradC4Params {
    const float* x; const void* dy; float* slab; float* bias_slab;      // dy: fp32, or bf16 (kernel template YB: MRDIS_DT_XF32_YBF16)
    int N, H, W, ldx, Co, lddy;
    int rowp, segs, R, splits;
    unsigned x_bytes, dy_bytes;
};

template <int NT, bool YB>
__global__ __launch_bounds__(256) void wgrad_c4_kernel(const WgradC4Params p) {
    constexpr int MT = 3, DYP = 16 * NT + 16, CO = 16 * NT;
    constexpr int YI = YB ? 4 : 8;                    // dy 16-byte items per thread and row: W * C / 1024 <= 8 fp32 quads (host), half as many bf16 octets
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                 // [4][rowp]
    float* dys = smem + 4 * p.rowp;                   // [W][DYP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int split = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = split / p.segs, oy0 = (split - n * p.segs) * p.R, oy1 = min(oy0 + p.R, p.H);
    const int W = p.W, WQ = W >> 6;                   // host: W % 64 == 0, W <= 256

    int acol[MT], aty[MT]; bool aok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l16;
        aok[mt] = m < 36;
        const int tap = aok[mt] ? m >> 2 : 0, ci = m & 3;
        aty[mt] = tap / 3;
        acol[mt] = (tap % 3) * 4 + ci + 4 * kq;
    }
    const int boff = kq * DYP + l16;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    for (int i = tid; i < 4 * p.rowp; i += 256) xs[i] = 0.f;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

    // staging: x row = W float4 (thread tid < W takes pixel tid); dy row = W * C / 4 float4: item = (pixel, quad), 4 NT quads per pixel
    u32x4 xr[2], yr[2][YI];
    auto load_set = [&](auto S_, int iy, int oy) {    // input row iy and dy row oy; nothing to fetch -> offsets out of range -> zeros
        constexpr int S = decltype(S_)::value;
        const bool xok = tid < W && iy <= oy1 && (unsigned)iy < (unsigned)p.H;
        xr[S] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(xok ? 4u * (unsigned)(((n * p.H + iy) * W + tid) * p.ldx) : S2_OOB), 0, 0);
        const bool on = oy >= 0 && oy < oy1;
        const unsigned ybase = (YB ? 2u : 4u) * (unsigned)((n * p.H + oy) * W * p.lddy);
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            const int idx = tid + 256 * j;            // valid while < W * 4 NT = 256 WQ NT (bf16: W * 2 NT octets)
            if (YB) {
                const int pix = idx / (2 * NT), q = idx - pix * (2 * NT);
                yr[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)((on && 2 * j < WQ * NT) ? ybase + 2u * (unsigned)(pix * p.lddy + 8 * q) : S2_OOB), 0, 0);
            } else {
                const int pix = idx / (4 * NT), q = idx - pix * (4 * NT);
                yr[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)((on && j < WQ * NT) ? ybase + 4u * (unsigned)(pix * p.lddy + 4 * q) : S2_OOB), 0, 0);
            }
        }
    };
    auto store_set = [&](auto S_, int iy, bool with_dy) {
        constexpr int S = decltype(S_)::value;
        if (tid < W) *reinterpret_cast<u32x4*>(xs + ((iy + 1) & 3) * p.rowp + (tid + 1) * 4) = xr[S];
        if (with_dy) {
#pragma unroll
            for (int j = 0; j < YI; ++j) {
                const int idx = tid + 256 * j;
                if (YB) {                             // eight bf16 -> eight fp32 (exact): the LDS image and the MFMAs are those of the fp32 form
                    const int pix = idx / (2 * NT), q = idx - pix * (2 * NT);
                    if (2 * j < WQ * NT) {
                        const u32x4 v = yr[S][j];
                        *reinterpret_cast<u32x4*>(dys + pix * DYP + 8 * q) = u32x4{v.x << 16, v.x & 0xffff0000u, v.y << 16, v.y & 0xffff0000u};
                        *reinterpret_cast<u32x4*>(dys + pix * DYP + 8 * q + 4) = u32x4{v.z << 16, v.z & 0xffff0000u, v.w << 16, v.w & 0xffff0000u};
                    }
                } else {
                    const int pix = idx / (4 * NT), q = idx - pix * (4 * NT);
                    if (j < WQ * NT) *reinterpret_cast<u32x4*>(dys + pix * DYP + 4 * q) = yr[S][j];
                }
            }
        }
    };
    const int nsteps = W >> 4;                        // a wave takes every fourth group of 4 pixels
    auto compute = [&](int oy) {
        int ao[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ao[mt] = ((oy + aty[mt]) & 3) * p.rowp + acol[mt];     // input row oy - 1 + ty -> slot (oy + ty) & 3
        float bvn[NT], avn[MT];
        auto read_ops = [&](int i) {
            const int ox0 = 4 * (wave + 4 * i);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bvn[nt] = dys[ox0 * DYP + boff + 16 * nt];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) avn[mt] = xs[ao[mt] + 4 * ox0];
        };
        read_ops(0);
        for (int i = 0; i < nsteps; ++i) {
            float bv[NT], av[MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { bv[nt] = bvn[nt]; bsum[nt] += bv[nt]; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = aok[mt] ? avn[mt] : 0.f;
            read_ops(i + 1 < nsteps ? i + 1 : i);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
    };
    auto step = [&](auto S_, int oy) {                // set S holds (input row oy + 2, dy row oy + 1)
        compute(oy);
        __syncthreads();
        store_set(S_, oy + 2, true);
        load_set(S_, oy + 4, oy + 3);
        __syncthreads();
    };
    // prologue: rows oy0 - 1, oy0, oy0 + 1 and dy row oy0 in LDS; sets hold (oy0 + 2, dy oy0 + 1) and (oy0 + 3, dy oy0 + 2)
    load_set(IC<0>{}, oy0 - 1, -1);
    load_set(IC<1>{}, oy0, oy0);
    __syncthreads();                                  // the zero fill
    store_set(IC<0>{}, oy0 - 1, false);
    store_set(IC<1>{}, oy0, true);
    load_set(IC<0>{}, oy0 + 1, -1);
    __syncthreads();
    store_set(IC<0>{}, oy0 + 1, false);
    load_set(IC<0>{}, oy0 + 2, oy0 + 1);
    load_set(IC<1>{}, oy0 + 3, oy0 + 2);
    __syncthreads();
    for (int oy = oy0; oy < oy1; oy += 2) {           // an odd row count runs one more row on an all-zero dy row
        step(IC<0>{}, oy);
        step(IC<1>{}, oy + 1);
    }

    float* red = smem;
    float* out = p.slab + (long long)split * 36 * CO;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NT + nt) * 256 + (4 * kq + r) * 16 + l16] = acc[mt][nt][r];
        __syncthreads();
        for (int e = tid; e < NT * 256; e += 256) {
            const int nt = e >> 8, rr = (e >> 4) & 15, cc = e & 15, m = mt * 16 + rr;
            const float v = (red[e] + red[NT * 256 + e]) + (red[2 * NT * 256 + e] + red[3 * NT * 256 + e]);
            if (m < 36) out[m * CO + nt * 16 + cc] = v;
        }
        __syncthreads();
    }
    if (p.bias_slab != nullptr) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float b = bsum[nt];
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            if (kq == 0) red[(wave * NT + nt) * 16 + l16] = b;
        }
        __syncthreads();
        if (tid < CO) {
            const int nt = tid >> 4, cc = tid & 15;
            p.bias_slab[(long long)split * CO + tid] = (red[nt * 16 + cc] + red[(NT + nt) * 16 + cc]) + (red[(2 * NT + nt) * 16 + cc] + red[(3 * NT + nt) * 16 + cc]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- 4 -> C, stride 1, bf16 dy
// The same weight (+ bias) gradient under `compute_dtype: bf16` (dy is a bf16 view: MRDIS_DT_XF32_YBF16) on the bf16 matrix pipe.  wgrad_c4_kernel<NT, true>
// widens dy to fp32 and IS the fp32 kernel: 384 fp32 MFMAs of 32 cycles per 256-pixel row, 41 us of matrix-pipe time on the 256x256 level against 30 us of
// memory time -- 90 us measured.  Here the reduction axis (the pixels of a row) is the K of v_mfma_f32_16x16x32_bf16, 32 pixels per MFMA:
//   * dy rows sit in LDS as they are in memory ([pixel][C] bf16, no widening); the B operand (8 consecutive pixels of one cout per lane) comes out of
//     two transposing reads, ds_read_b64_tr_b16 (a 16-lane group reads 4 pixel rows x 16 couts and lane i receives cout i of the 4 pixels);
//   * the fp32 anatomy map is split into THREE bf16 terms on its way into LDS, x = hi + mid + lo (3 x 8 mantissa bits: exact), stored PLANAR --
//     [ring slot][term][channel][pixel + 1] -- so that the A operand of patch row (tap, ci) is 8 consecutive bf16 of one plane, an aligned 16-byte LDS read
//     + the next dword, shifted by the tap column tx in registers (v_alignbit_b32);  the products are those of the fp32 form, exactly, only the
//     order of the fp32 sum differs;
//   * patch row 36 (unused in the third 16-row tile) is a row of ones: its D row is the column sum of dy, the bias gradient, for free.
// 9 NT bf16 MFMAs of 16 cycles per 32 pixels; slab format, row split across the four waves, final cross-wave sum and the slab reduction are wgrad_c4_kernel's.
typedef __bf16 wc_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wc_s16x4 __attribute__((ext_vector_type(4)));

// SWAP: the C -> 4 layer (ana_dec.output, MRDIS_DT_XBF16_YF32: x a bf16 view of C channels, dy fp32 with 4) is the same sum with the roles exchanged,
//   dW[tap][ci][co] = sum_p x[p + off(tap)][ci] dy[p][co] = sum_p' dy[p' + off(8 - tap)][co] x[p'][ci]:
// the four-channel fp32 tensor (`p.x` here = dy) is the one that is shifted and split, the bf16 rows (`p.dy` here = x) are the ones read transposed; the
// slab is written in the layer's own [tap][ci][co] order (tap reversed, the two channel axes exchanged) so that the shared slab reduction needs no other
// form, and the bias gradient (column sums of the FOUR-channel tensor over the workgroup's own rows) comes from the staging registers.
template <int NT, int YI = 4, bool SWAP = false>
__global__ __launch_bounds__(256) void wgrad_c4b_kernel(const WgradC4Params p) {
    constexpr int MT = 3, CO = 16 * NT;
    constexpr int DYPB = 2 * CO + (CO == 32 ? 0 : 16);            // bytes of a dy pixel row in LDS: the four rows of a transposing read fall on distinct banks
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int W = p.W, PL = W + 16;                    // plane length (bf16 elements): pixel b at index b + 1, zero columns at 0 and W + 1
    __bf16* xs = reinterpret_cast<__bf16*>(smem_b);   // [4 slots][3 terms][4 ci][PL]
    unsigned char* dys = smem_b + (size_t)2 * 48 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int split = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = split / p.segs, oy0 = (split - n * p.segs) * p.R, oy1 = min(oy0 + p.R, p.H);

    // patch row m = 16 mt + l16 = 4 tap + ci of this lane in each row tile; m = 36: ones; m > 36: zeros
    int aty[MT], aoff[MT], atx[MT]; bool aok[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l16;
        aok[mt] = m < 36;
        const int tap = aok[mt] ? m >> 2 : 0, ci = m & 3;
        aty[mt] = tap / 3;
        aoff[mt] = ci * PL + 8 * kq;                   // + (slot * 3 + term) * 4 * PL + b0: 16-byte aligned; the tap column tx = tap % 3 shifts in registers
        atx[mt] = tap % 3;
    }
    const bool ones_row = l16 == 4;                    // (in row tile 2)
    const int q4 = l16 >> 2, p4 = l16 & 3;
    const int boff = (8 * kq + q4) * DYPB + 8 * p4;    // transposing read: this lane supplies row q4, columns 4 p4 .. + 3 of its group's block
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    for (int i = tid; i < 48 * PL / 2; i += 256) reinterpret_cast<unsigned*>(xs)[i] = 0u;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    float bs4[4] = {0.f, 0.f, 0.f, 0.f};              // SWAP: this thread's share of the bias gradient
    u32x4 xr[2], yr[2][YI];
    auto load_set = [&](auto S_, int iy, int oy) {    // input row iy and dy row oy; nothing to fetch -> offsets out of range -> zeros
        constexpr int S = decltype(S_)::value;
        const bool xok = tid < W && iy <= oy1 && (unsigned)iy < (unsigned)p.H;
        xr[S] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(xok ? 4u * (unsigned)(((n * p.H + iy) * W + tid) * p.ldx) : S2_OOB), 0, 0);
        const bool on = oy >= 0 && oy < oy1;
        const unsigned ybase = 2u * (unsigned)((n * p.H + oy) * W * p.lddy);
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            const int idx = tid + 256 * j;            // 16-byte piece (pixel, q) of the row: valid while < W * 2 NT
            const int pix = idx / (2 * NT), q = idx - pix * (2 * NT);
            yr[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)((on && idx < W * 2 * NT) ? ybase + 2u * (unsigned)(pix * p.lddy + 8 * q) : S2_OOB), 0, 0);
        }
    };
    auto store_set = [&](auto S_, int iy, bool with_dy) {
        constexpr int S = decltype(S_)::value;
        if (tid < W) {
            __bf16* dst = xs + (size_t)((iy + 1) & 3) * 12 * PL + tid + 1;
            const float xv[4] = {__uint_as_float(xr[S].x), __uint_as_float(xr[S].y), __uint_as_float(xr[S].z), __uint_as_float(xr[S].w)};
            if (SWAP && iy >= oy0 && iy < oy1) { bs4[0] += xv[0]; bs4[1] += xv[1]; bs4[2] += xv[2]; bs4[3] += xv[3]; }
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
                const __bf16 hi = (__bf16)xv[ci]; const float r1 = xv[ci] - (float)hi;
                const __bf16 mid = (__bf16)r1; const __bf16 lo = (__bf16)(r1 - (float)mid);
                dst[ci * PL] = hi; dst[(4 + ci) * PL] = mid; dst[(8 + ci) * PL] = lo;
            }
        }
        if (with_dy) {
#pragma unroll
            for (int j = 0; j < YI; ++j) {
                const int idx = tid + 256 * j;
                const int pix = idx / (2 * NT), q = idx - pix * (2 * NT);
                if (idx < W * 2 * NT) *reinterpret_cast<u32x4*>(dys + pix * DYPB + 16 * q) = yr[S][j];
            }
        }
    };
    const int nsteps = W >> 5;                         // 32-pixel k-steps of a row; wave w takes steps w, w + 4, ..
    auto compute = [&](int oy) {
        const __bf16* arow[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) arow[mt] = xs + (size_t)((oy + aty[mt]) & 3) * 12 * PL + aoff[mt];     // input row oy - 1 + ty -> slot (oy + ty) & 3
        for (int i = wave; i < nsteps; i += 4) {
            const int b0 = 32 * i;
            wc_bf16x8 bq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                union { wc_bf16x8 v; wc_s16x4 h[2]; } u;
                const unsigned char* src = dys + b0 * DYPB + boff + 32 * nt;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wc_s16x4 __attribute__((address_space(3)))*)(src));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wc_s16x4 __attribute__((address_space(3)))*)(src + 4 * DYPB));
                bq[nt] = u.v;
            }
#pragma unroll
            for (int term = 2; term >= 0; --term) {    // smallest terms first
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // ten consecutive bf16 of the plane from the aligned index, as five dwords; the lane's eight start tx elements in (a misaligned
                    // 16-byte LDS read does it in one instruction, and costs ~500 cycles: 84 us per launch against 38)
                    const __bf16* src = arow[mt] + term * 4 * PL + b0;
                    const u32x4 d = *reinterpret_cast<const u32x4*>(src);
                    const unsigned d4 = *reinterpret_cast<const unsigned*>(src + 8);
                    const unsigned dd[5] = {d.x, d.y, d.z, d.w, d4};
                    const unsigned sh = (atx[mt] & 1) ? 16u : 0u;
                    u32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned r = __builtin_amdgcn_alignbit(dd[j + 1], dd[j], sh);
                        o[j] = atx[mt] == 2 ? dd[j + 1] : r;
                    }
                    wc_bf16x8 a = __builtin_bit_cast(wc_bf16x8, o);
                    if (!aok[mt]) {
                        const __bf16 fill = (mt == 2 && ones_row && term == 0) ? (__bf16)1.f : (__bf16)0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[j] = fill;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    };
    auto step = [&](auto S_, int oy) {                // set S holds (input row oy + 2, dy row oy + 1)
        compute(oy);
        __syncthreads();
        store_set(S_, oy + 2, true);
        load_set(S_, oy + 4, oy + 3);
        __syncthreads();
    };
    // prologue: rows oy0 - 1, oy0, oy0 + 1 and dy row oy0 in LDS; sets hold (oy0 + 2, dy oy0 + 1) and (oy0 + 3, dy oy0 + 2)
    load_set(IC<0>{}, oy0 - 1, -1);
    load_set(IC<1>{}, oy0, oy0);
    __syncthreads();                                  // the zero fill
    store_set(IC<0>{}, oy0 - 1, false);
    store_set(IC<1>{}, oy0, true);
    load_set(IC<0>{}, oy0 + 1, -1);
    __syncthreads();
    store_set(IC<0>{}, oy0 + 1, false);
    load_set(IC<0>{}, oy0 + 2, oy0 + 1);
    load_set(IC<1>{}, oy0 + 3, oy0 + 2);
    __syncthreads();
    for (int oy = oy0; oy < oy1; oy += 2) {           // an odd row count runs one more row on an all-zero dy row
        step(IC<0>{}, oy);
        step(IC<1>{}, oy + 1);
    }

    float* red = reinterpret_cast<float*>(smem_b);
    float* out = p.slab + (long long)split * 36 * CO;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NT + nt) * 256 + (4 * kq + r) * 16 + l16] = acc[mt][nt][r];
        __syncthreads();
        for (int e = tid; e < NT * 256; e += 256) {
            const int nt = e >> 8, rr = (e >> 4) & 15, cc = e & 15, m = mt * 16 + rr;
            const float v = (red[e] + red[NT * 256 + e]) + (red[2 * NT * 256 + e] + red[3 * NT * 256 + e]);
            if (SWAP) { if (m < 36) out[((8 - (m >> 2)) * CO + nt * 16 + cc) * 4 + (m & 3)] = v; }
            else if (m < 36) out[m * CO + nt * 16 + cc] = v;
            else if (m == 36 && p.bias_slab != nullptr) p.bias_slab[(long long)split * CO + nt * 16 + cc] = v;
        }
        __syncthreads();
    }
    if (SWAP && p.bias_slab != nullptr) {
#pragma unroll
        for (int c = 0; c < 4; ++c) red[c * 256 + tid] = bs4[c];
        __syncthreads();
        if (tid < 4) {
            float t = 0.f;
            for (int k = 0; k < 256; ++k) t += red[tid * 256 + k];
            p.bias_slab[(long long)split * 4 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- data gradient
// dx of the same stride-2 layers (the second encoder pass back-propagates into the generated images): a window-free GEMM per dy pixel q,
//     Z[m = (tap, ci)][q] = sum_co w[tap][co][ci] dy[q][co]            (112 = 7 tiles | 63 rows, K = Co: no padding waste),
// and a gather  dx[2 q + tap - 1][ci] += Z[(tap, ci)][q]:  every dx pixel takes the 2 x 2 taps of its parity class.  As in mrdis_co4.hip
// the dy operand goes from global memory straight into MFMA registers, the filter stays in registers, Z of one dy row goes to LDS
// ([q + 1][M + pad], conflict-free 16-byte writes) and thread ix adds its taps in fixed order while the workgroup streams down the dy
// rows of one image: dy row r completes dx rows 2 r - 1 (tap rows 2 of r - 1 and 0 of r) and 2 r (tap rows 3 of r - 1 and 1 of r).
struct DgradS2Params {
    const float* dy; const float* w; float* dx;
    int N, H, W, Ci, lddx, Co, lddy, Hout, Wout;
    int R, segs, M, MP;
    unsigned dy_bytes;
};

template <int KS, int HALVES>
__global__ __launch_bounds__(256, 2) void dgrad_s2_kernel(const DgradS2Params p) {
    constexpr int MT = (KS * KS * 7 + 15) / 16, KSTEPS = 4 * HALVES, CO = 16 * HALVES;
    extern __shared__ __attribute__((aligned(16))) float zs[];      // [Wout + 2][MP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int Ci = p.Ci, MP = p.MP;
    const int b = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = b / p.segs, r0 = (b - n * p.segs) * p.R, r1 = min(r0 + p.R, p.Hout);
    // A = filter: a[ks][mt] = w[tap][co = 16 h + 4 kq + j][ci], row m = 16 mt + l16 = tap * Ci + ci, ks = 4 h + j
    float a[KSTEPS][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + l16;
        const bool ok = m < p.M;
        const int tap = ok ? m / Ci : 0, ci = ok ? m - tap * Ci : 0;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) a[ks][mt] = ok ? p.w[(tap * CO + 16 * (ks >> 2) + 4 * kq + (ks & 3)) * Ci + ci] : 0.f;
    }
    for (int i = tid; i < (p.Wout + 2) * MP; i += 256) zs[i] = 0.f;    // rows q = -1 and q = Wout stay zero
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const int tpw = p.Wout >> 6;                      // 16-pixel tiles per wave and row (host: Wout % 64 == 0, <= 2)
    u32x4 yr[2][HALVES];
    auto load_tile = [&](int t, int r) {
        const int q = 16 * (wave + 4 * t) + l16;
        const bool ok = t < tpw && (unsigned)r < (unsigned)p.Hout && r >= r0 - 1 && r <= r1;
        const unsigned base = 4u * (unsigned)(((n * p.Hout + r) * p.Wout + q) * p.lddy + 4 * kq);
#pragma unroll
        for (int h = 0; h < HALVES; ++h) yr[t][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)(ok ? base + 64u * h : S2_OOB), 0, 0);
    };
    // thread ix = tid: its column's taps.  ix even (2 b): tx = 1 -> q = b, tx = 3 -> q = b - 1;  ix odd (2 b + 1): tx = 0 -> q = b + 1, tx = 2 -> q = b
    const int ix = tid, bcol = ix >> 1;
    const int txa = (ix & 1) ? 0 : 1, qa = (ix & 1) ? bcol + 1 : bcol;          // first tap column and its dy pixel
    const int txb = txa + 2, qb = (ix & 1) ? bcol : bcol - 1;                    // second (absent for 3x3 taps when tx = 3)
    const bool has_b = txb < KS;
    float runE[7], runO[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) { runE[c] = 0.f; runO[c] = 0.f; }
    auto tap_row = [&](int ty, float (&g)[7]) {       // sum over this column's taps of tap row ty, fixed order
        const float* za = zs + (qa + 1) * MP + (ty * KS + txa) * Ci;
        const float* zb = zs + (qb + 1) * MP + (ty * KS + txb) * Ci;
#pragma unroll
        for (int c = 0; c < 7; ++c) g[c] = c < Ci ? za[c] + (has_b ? zb[c] : 0.f) : 0.f;
    };

#pragma unroll
    for (int t = 0; t < 2; ++t) load_tile(t, r0 - 1);
    __syncthreads();
    for (int r = r0 - 1; r <= r1; ++r) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t < tpw) {
                f32x4 acc[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < HALVES; ++h) {
                    const float yv[4] = {__uint_as_float(yr[t][h].x), __uint_as_float(yr[t][h].y), __uint_as_float(yr[t][h].z), __uint_as_float(yr[t][h].w)};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * h + j][mt], yv[j], acc[mt], 0, 0, 0);
                }
                float* zd = zs + (16 * (wave + 4 * t) + l16 + 1) * MP + 4 * kq;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    if (16 * mt < p.M) *reinterpret_cast<f32x4*>(zd + 16 * mt) = acc[mt];      // (row tiles beyond taps * Ci do not exist in the MP-float pixel row)
            }
            load_tile(t, r + 1);
        }
        __syncthreads();
        if (ix < p.W) {
            float g0[7], g1[7], g2[7], g3[7];
            tap_row(0, g0); tap_row(1, g1); tap_row(2, g2);
            if (KS == 4) tap_row(3, g3);
            // dx row 2 r - 1 = tap row 2 of dy row r - 1 (runO) + tap row 0 of dy row r;  dx row 2 r = tap row 3 of r - 1 (runE) + tap row 1 of r
            if (r - 1 >= r0 && r - 1 < r1) {
                float* d = p.dx + ((long long)(n * p.H + 2 * r - 1) * p.W + ix) * p.lddx;
#pragma unroll
                for (int c = 0; c < 7; ++c) if (c < Ci) d[c] = runO[c] + g0[c];
            }
            if (r >= r0 && r < r1) {
                float* d = p.dx + ((long long)(n * p.H + 2 * r) * p.W + ix) * p.lddx;
#pragma unroll
                for (int c = 0; c < 7; ++c) if (c < Ci) d[c] = runE[c] + g1[c];
            }
#pragma unroll
            for (int c = 0; c < 7; ++c) { runO[c] = g2[c]; runE[c] = KS == 4 ? g3[c] : 0.f; }
        }
        __syncthreads();
    }
}

// dw[i] = sum_k slab[k][i] (i < total), dbias[co] (+)= sum_k bslab[k][co]: the fixed-order slab reduction shared with mrdis_pointwise.hip
int mrdis_launch_slab_reduce(const float* slab, float* dw, int total, int Co, int nslab, const float* bslab, float* dbias, int accumulate_bias,
                             hipStream_t s) {
    const int nout = total + (dbias ? Co : 0);
    MRDIS_LAUNCH(wgrad_s2_reduce_kernel, dim3(mrdis_cdiv(nout, 32)), dim3(32, 32), 0, s, slab, dw, total, Co, nslab, bslab, dbias, accumulate_bias, SlabPad{0, 0, 0, 0});
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
// the same with dw = [T][rout][cout] around slabs of [T][rin][cin]
static int launch_slab_reduce_padded(const float* slab, float* dw, int T, int rin, int cin, int rout, int cout, int Co, int nslab, const float* bslab,
                                     float* dbias, int accumulate_bias, hipStream_t s) {
    const int total = T * rout * cout, nout = total + (dbias ? Co : 0);
    MRDIS_LAUNCH(wgrad_s2_reduce_kernel, dim3(mrdis_cdiv(nout, 32)), dim3(32, 32), 0, s, slab, dw, total, Co, nslab, bslab, dbias, accumulate_bias,
                 SlabPad{rin, cin, rout, cout});
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

static bool plan_wgrad_s2(WgradS2Params& p, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad) {
    if (stride != 2 || pad != 1 || kh != kw || (kh != 3 && kh != 4)) return false;
    if (Ci < 1 || Ci > 7 || (Co != 16 && Co != 32)) return false;
    if ((H & 1) || W % 32 != 0 || W * Ci > 256 * S2_XJ) return false;        // Wout % 16 == 0; a row's items fit the staging registers
    if ((long long)N * H * W < 100000 || mrdis_opt(MRDIS_OPT_NOW16)) return false;     // small maps: the generic kernel's slabs are cheaper
    p = WgradS2Params{};
    p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co; p.Hout = H / 2; p.Wout = W / 2;
    p.M = kh * kw * Ci;
    // x-row pitch: >= (W + 2) pixels, and = kw * Ci mod 64 so that the (tap, ci) sequence runs on across tap rows bank-wise
    int rowp = (W + 2) * Ci;
    while ((rowp & 63) != ((kw * Ci) & 63)) ++rowp;
    p.rowp = rowp;
    // ~512 workgroups (2 per CU), each a run of consecutive output rows of one image (consecutive rows share an input-row pair)
    int segs = mrdis_cdiv(512, N);
    if (segs > p.Hout) segs = p.Hout;
    p.R = mrdis_cdiv(p.Hout, segs);
    p.segs = mrdis_cdiv(p.Hout, p.R);
    p.splits = N * p.segs;
    return true;
}

size_t mrdis_wgrad_s2_workspace(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad) {
    WgradS2Params p;
    if (!plan_wgrad_s2(p, N, H, W, Ci, Co, kh, kw, stride, pad)) return 0;
    return sizeof(float) * ((size_t)p.splits * p.M * Co + (size_t)p.splits * Co) + 256;
}

// returns MRDIS_EUNSUPPORTED when the layer is outside what this kernel covers
int mrdis_run_wgrad_s2(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                       size_t workspace_bytes, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                       int accumulate_bias, hipStream_t s) {
    WgradS2Params p;
    if (!plan_wgrad_s2(p, N, H, W, Ci, Co, kh, kw, stride, pad)) return MRDIS_EUNSUPPORTED;
    if (lddy % 4 != 0 || (((uintptr_t)dy) & 15) != 0 || (((uintptr_t)x) & 3) != 0) return MRDIS_EUNSUPPORTED;
    const long long xb = 4LL * (((long long)N * H * W - 1) * ldx + Ci), yb = 4LL * (((long long)N * p.Hout * p.Wout - 1) * lddy + Co);
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_wgrad_s2_workspace(N, H, W, Ci, Co, kh, kw, stride, pad)) return MRDIS_EUNSUPPORTED;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy; p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + (size_t)p.splits * p.M * Co : nullptr;
    const int NT = Co / 16, DYP = NT == 2 ? 48 : 16;
    size_t lds = sizeof(float) * ((size_t)4 * p.rowp + (size_t)p.Wout * DYP);
    const size_t red = sizeof(float) * (size_t)(4 * NT * 256);
    if (lds < red) lds = red;
    if (lds > 64 * 1024) return MRDIS_EUNSUPPORTED;
    if (kh == 4 && NT == 2) MRDIS_LAUNCH((wgrad_s2_kernel<4, 2>), dim3(p.splits), dim3(256), lds, s, p);
    else if (kh == 4) MRDIS_LAUNCH((wgrad_s2_kernel<4, 1>), dim3(p.splits), dim3(256), lds, s, p);
    else if (NT == 2) MRDIS_LAUNCH((wgrad_s2_kernel<3, 2>), dim3(p.splits), dim3(256), lds, s, p);
    else MRDIS_LAUNCH((wgrad_s2_kernel<3, 1>), dim3(p.splits), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    const int total = p.M * Co;
    return mrdis_launch_slab_reduce(p.slab, dw_tck, total, Co, p.splits, p.bias_slab, dbias, accumulate_bias, s);
}

// forward of the same layers; MRDIS_EUNSUPPORTED outside what the kernel covers
int mrdis_run_conv_s2_fwd(const float* x, int ldx, const float* w_tck, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co,
                          int kh, int kw, int stride, int pad, int lrelu, hipStream_t s) {
    WgradS2Params q;
    if (!plan_wgrad_s2(q, N, H, W, Ci, Co, kh, kw, stride, pad)) return MRDIS_EUNSUPPORTED;
    if (ldy % 4 != 0 || (((uintptr_t)y) & 15) != 0 || (((uintptr_t)x) & 3) != 0) return MRDIS_EUNSUPPORTED;
    const long long xb = 4LL * (((long long)N * H * W - 1) * ldx + Ci);
    if (xb >= 0x7fffffffLL || (long long)N * q.Hout * q.Wout * ldy >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    ConvS2Params p{};
    p.x = x; p.w = w_tck; p.bias = bias; p.y = y; p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.ldx = ldx; p.Co = Co; p.ldy = ldy;
    p.Hout = q.Hout; p.Wout = q.Wout; p.lrelu = lrelu; p.rowp = q.rowp; p.segs = q.segs; p.R = q.R; p.M = q.M; p.x_bytes = (unsigned)xb;
    const size_t lds = sizeof(float) * (size_t)4 * p.rowp;
    const int NT = Co / 16;
    if (kh == 4 && NT == 2) MRDIS_LAUNCH((conv_s2_fwd_kernel<4, 2>), dim3(q.splits), dim3(256), lds, s, p);
    else if (kh == 4) MRDIS_LAUNCH((conv_s2_fwd_kernel<4, 1>), dim3(q.splits), dim3(256), lds, s, p);
    else if (NT == 2) MRDIS_LAUNCH((conv_s2_fwd_kernel<3, 2>), dim3(q.splits), dim3(256), lds, s, p);
    else MRDIS_LAUNCH((conv_s2_fwd_kernel<3, 1>), dim3(q.splits), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

static bool plan_wgrad_c4(WgradC4Params& p, int N, int H, int W, int Ci, int Co) {
    if (Ci != 4 || (Co != 32 && Co != 64 && Co != 128) || (W != 64 && W != 128 && W != 256)) return false;
    if ((long long)W * Co > 256 * 32 || (long long)N * H * W < 100000 || mrdis_opt(MRDIS_OPT_NOW16)) return false;    // a dy row fits the staging registers
    p = WgradC4Params{};
    p.N = N; p.H = H; p.W = W; p.Co = Co;
    int rowp = (W + 2) * 4;
    while ((rowp & 63) != 12) ++rowp;                  // the (tap, ci) sequence of a tap row (12 words) runs on into the next ring slot bank-wise
    p.rowp = rowp;
    int segs = mrdis_cdiv(512, N);
    if (segs > H / 2) segs = H / 2 > 0 ? H / 2 : 1;
    p.R = mrdis_cdiv(H, segs); p.segs = mrdis_cdiv(H, p.R);
    p.splits = N * p.segs;
    return true;
}

size_t mrdis_wgrad_c4_workspace(int N, int H, int W, int Ci, int Co) {
    WgradC4Params p;
    if (!plan_wgrad_c4(p, N, H, W, Ci, Co)) return 0;
    return sizeof(float) * ((size_t)p.splits * 36 * Co + (size_t)p.splits * Co) + 256;
}

// weight gradient of a 4 -> C 3x3 s1 p1 layer; MRDIS_EUNSUPPORTED outside what the kernel covers
int mrdis_run_wgrad_c4(const float* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                       int N, int H, int W, int Ci, int Co, int accumulate_bias, int dy_bf16, hipStream_t s, int pad16) {
    WgradC4Params p;
    if (!plan_wgrad_c4(p, N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    if (ldx % 4 != 0 || lddy % (dy_bf16 ? 8 : 4) != 0 || ((((uintptr_t)x) | ((uintptr_t)dy)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    const long long xb = 4LL * (((long long)N * H * W - 1) * ldx + 4), yb = (dy_bf16 ? 2LL : 4LL) * (((long long)N * H * W - 1) * lddy + Co);
    if (xb >= 0x7fffffffLL || yb >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_wgrad_c4_workspace(N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy; p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)yb;
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + (size_t)p.splits * 36 * Co : nullptr;
    const int NT = Co / 16;
    const size_t red = sizeof(float) * (size_t)(4 * NT * 256);
    if (dy_bf16 && mrdis_opt(MRDIS_OPT_MODE) != 3011) {            // the bf16 matrix pipe (debug_mode 3011: the widening fp32 form below)
        size_t ldsb = (size_t)2 * 48 * (W + 16) + (size_t)W * (2 * Co + (Co == 32 ? 0 : 16));
        if (ldsb < red) ldsb = red;
        if (ldsb > 64 * 1024) return MRDIS_EUNSUPPORTED;
        if (NT == 2) MRDIS_LAUNCH((wgrad_c4b_kernel<2>), dim3(p.splits), dim3(256), ldsb, s, p);
        else if (NT == 4) MRDIS_LAUNCH((wgrad_c4b_kernel<4>), dim3(p.splits), dim3(256), ldsb, s, p);
        else MRDIS_LAUNCH((wgrad_c4b_kernel<8>), dim3(p.splits), dim3(256), ldsb, s, p);
        MRDIS_CHECK_LAUNCH();
        if (pad16) return launch_slab_reduce_padded(p.slab, dw_tck, 9, 4, Co, 16, Co, Co, p.splits, p.bias_slab, dbias, accumulate_bias, s);
        return mrdis_launch_slab_reduce(p.slab, dw_tck, 36 * Co, Co, p.splits, p.bias_slab, dbias, accumulate_bias, s);
    }
    if (pad16) return MRDIS_EUNSUPPORTED;
    size_t lds = sizeof(float) * ((size_t)4 * p.rowp + (size_t)W * (Co + 16));
    if (lds < red) lds = red;
    if (lds > 72 * 1024) return MRDIS_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {          // > 64 KB of dynamic LDS needs the opt-in (4 -> 32 at W = 256: 65.7 KB)
        if (hipFuncSetAttribute((const void*)wgrad_c4_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_c4_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_c4_kernel<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_c4_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_c4_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_c4_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess)
            return MRDIS_EUNSUPPORTED;
        attr_set = true;
    }
#define WC4_LAUNCH(nt) { if (dy_bf16) MRDIS_LAUNCH((wgrad_c4_kernel<nt, true>), dim3(p.splits), dim3(256), lds, s, p); \
                         else MRDIS_LAUNCH((wgrad_c4_kernel<nt, false>), dim3(p.splits), dim3(256), lds, s, p); }
    if (NT == 2) WC4_LAUNCH(2) else if (NT == 4) WC4_LAUNCH(4) else WC4_LAUNCH(8)
#undef WC4_LAUNCH
    MRDIS_CHECK_LAUNCH();
    return mrdis_launch_slab_reduce(p.slab, dw_tck, 36 * Co, Co, p.splits, p.bias_slab, dbias, accumulate_bias, s);
}

// weight gradient of a C -> 4 3x3 s1 p1 layer on a bf16 input view and an fp32 gradient (MRDIS_DT_XBF16_YF32): wgrad_c4b_kernel<.., SWAP>
static bool plan_wgrad_co4b(WgradC4Params& p, int N, int H, int W, int Ci, int Co) {
    if (Co != 4 || (Ci != 32 && Ci != 64) || (W != 64 && W != 128 && W != 256)) return false;
    if ((long long)W * Ci > 256 * 64 || (long long)N * H * W < 100000 || mrdis_opt(MRDIS_OPT_NOW16)) return false;    // a row of x fits the staging registers
    p = WgradC4Params{};
    p.N = N; p.H = H; p.W = W; p.Co = Ci;              // (the kernel's "Co" is the wide tensor's channel count)
    int segs = mrdis_cdiv(512, N);
    if (segs > H / 2) segs = H / 2 > 0 ? H / 2 : 1;
    p.R = mrdis_cdiv(H, segs); p.segs = mrdis_cdiv(H, p.R);
    p.splits = N * p.segs;
    return true;
}
size_t mrdis_wgrad_co4b_workspace(int N, int H, int W, int Ci, int Co) {
    WgradC4Params p;
    if (!plan_wgrad_co4b(p, N, H, W, Ci, Co)) return 0;
    return sizeof(float) * ((size_t)p.splits * 36 * Ci + (size_t)p.splits * 4) + 256;
}
int mrdis_run_wgrad_co4b(const void* x_bf16, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                         int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s, int pad16) {
    WgradC4Params p;
    if (!plan_wgrad_co4b(p, N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    if (lddy % 4 != 0 || ldx % 8 != 0 || ((((uintptr_t)x_bf16) | ((uintptr_t)dy)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    const long long fb = 4LL * (((long long)N * H * W - 1) * lddy + 4), wb = 2LL * (((long long)N * H * W - 1) * ldx + Ci);
    if (fb >= 0x7fffffffLL || wb >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_wgrad_co4b_workspace(N, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    p.x = dy; p.ldx = lddy; p.x_bytes = (unsigned)fb;               // the four-channel fp32 tensor
    p.dy = x_bf16; p.lddy = ldx; p.dy_bytes = (unsigned)wb;         // the wide bf16 rows
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + (size_t)p.splits * 36 * Ci : nullptr;
    const int NT = Ci / 16, pieces = W * 2 * NT / 256;              // 16-byte pieces of a wide row per thread: 1 .. 8
    size_t lds = (size_t)2 * 48 * (W + 16) + (size_t)W * (2 * Ci + (Ci == 32 ? 0 : 16));
    const size_t red = sizeof(float) * (size_t)(4 * NT * 256);
    if (lds < red) lds = red;
    if (lds > 64 * 1024) return MRDIS_EUNSUPPORTED;
    if (NT == 2 && pieces <= 4) MRDIS_LAUNCH((wgrad_c4b_kernel<2, 4, true>), dim3(p.splits), dim3(256), lds, s, p);
    else if (NT == 4 && pieces <= 4) MRDIS_LAUNCH((wgrad_c4b_kernel<4, 4, true>), dim3(p.splits), dim3(256), lds, s, p);
    else if (NT == 4 && pieces <= 8) MRDIS_LAUNCH((wgrad_c4b_kernel<4, 8, true>), dim3(p.splits), dim3(256), lds, s, p);
    else return MRDIS_EUNSUPPORTED;
    MRDIS_CHECK_LAUNCH();
    if (pad16) return launch_slab_reduce_padded(p.slab, dw_tck, 9, Ci, 4, Ci, 16, 4, p.splits, p.bias_slab, dbias, accumulate_bias, s);
    return mrdis_launch_slab_reduce(p.slab, dw_tck, 36 * Ci, 4, p.splits, p.bias_slab, dbias, accumulate_bias, s);
}

// data gradient of the same layers (dy (N, H/2, W/2, Co) -> dx (N, H, W, Ci)); MRDIS_EUNSUPPORTED outside what the kernel covers
int mrdis_run_dgrad_s2(const float* dy, int lddy, const float* w_tkc, float* dx, int lddx, int N, int H, int W, int Ci, int Co,
                       int kh, int kw, int stride, int pad, hipStream_t s) {
    if (stride != 2 || pad != 1 || kh != kw || (kh != 3 && kh != 4) || Ci < 1 || Ci > 7 || (Co != 16 && Co != 32)) return MRDIS_EUNSUPPORTED;
    if ((H & 1) || (W != 128 && W != 256) || lddy % 4 != 0 || (((uintptr_t)dy) & 15) != 0 || mrdis_opt(MRDIS_OPT_NOW16)) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W < 100000) return MRDIS_EUNSUPPORTED;
    DgradS2Params p{};
    p.dy = dy; p.w = w_tkc; p.dx = dx; p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.lddx = lddx; p.Co = Co; p.lddy = lddy;
    p.Hout = H / 2; p.Wout = W / 2; p.M = kh * kw * Ci;
    const int mrows = ((p.M + 15) / 16) * 16;
    p.MP = mrows + 4;                                  // (MP / 4 odd: the 16 pixels of a tile write to 16 distinct 4-bank groups)
    const long long yb = 4LL * (((long long)N * p.Hout * p.Wout - 1) * lddy + Co);
    if (yb >= 0x7fffffffLL || (long long)N * H * W * lddx >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.dy_bytes = (unsigned)yb;
    int segs = mrdis_cdiv(512, N);
    if (segs > p.Hout / 2) segs = p.Hout / 2 > 0 ? p.Hout / 2 : 1;
    p.R = mrdis_cdiv(p.Hout, segs); p.segs = mrdis_cdiv(p.Hout, p.R);
    const size_t lds = sizeof(float) * (size_t)(p.Wout + 2) * p.MP;
    if (lds > 72 * 1024) return MRDIS_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)dgrad_s2_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)dgrad_s2_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)dgrad_s2_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)dgrad_s2_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) != hipSuccess)
            return MRDIS_EUNSUPPORTED;
        attr_set = true;
    }
    const dim3 grid(N * p.segs), block(256);
    if (kh == 4 && Co == 32) MRDIS_LAUNCH((dgrad_s2_kernel<4, 2>), grid, block, lds, s, p);
    else if (kh == 4) MRDIS_LAUNCH((dgrad_s2_kernel<4, 1>), grid, block, lds, s, p);
    else if (Co == 32) MRDIS_LAUNCH((dgrad_s2_kernel<3, 2>), grid, block, lds, s, p);
    else MRDIS_LAUNCH((dgrad_s2_kernel<3, 1>), grid, block, lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
