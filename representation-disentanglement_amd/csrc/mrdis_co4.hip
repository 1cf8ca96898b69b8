// mrdis_co4.hip -- 3x3 / stride 1 / pad 1 convolution with FOUR output channels and 32 / 64 input channels, fp32:
// ana_dec.output (64 -> 4 at full resolution, model.py:2260) and the data gradients of the SPADE `si_layers` (C -> 4, model.py:2436).
// A 16-wide MFMA tile over 4 couts is 75 % padding and the packed-FMA kernel that ran these layers (tapconv16_kernel<., THIN4>) is
// LDS-issue bound at 2.4 TB/s.  Here the product is split into a GEMM that needs no window and a gather that needs no multiplies:
//     Z[tap][q][co] = sum_ci x[q][ci] w[tap][ci][co]        36 = 9 taps x 4 couts rows per INPUT pixel q: three 16-row MFMA tiles (75 % used),
//     y[p][co]      = bias[co] + sum_tap Z[tap][p + off(tap)][co]
// * the pixel operand of the GEMM is the unshifted input: it goes from global memory straight into MFMA registers (one 16-byte load per
//   16 channels and lane), no LDS staging, no halo; the filter (A operand, rows ordered co-major inside a tile) stays in registers;
// * D[row][pixel] leaves every lane (pixel, kq = co) with the values of 4 taps of its cout: nine conflict-free ds_write_b32 per tile put
//   Z[tap][pixel][co] into LDS (36 KB per 256-pixel row);
// * a workgroup streams down the rows of one image: after the Z of input row r is in LDS, thread p adds the three taps of each tap row
//   (fixed order) -- tap row 0 starts output row r + 1, tap row 1 continues output row r, tap row 2 completes output row r - 1, which is
//   stored; the two running rows live in the thread's registers.  Two barriers per row; next row's pixel operands are in flight in the
//   registers of the tile that was just multiplied.
#include "mrdis_common.h"

struct Co4Params {
    const void* x; const float* w; const float* bias; float* y;       // x: fp32, or bf16 (template XB: MRDIS_DT_XBF16_YF32 forward / MRDIS_DT_XF32_YBF16 data gradient)
    int N, H, W, Ci, ldx, ldy, flip, lrelu;
    int wld;                  // filter row length: 4, or 16 for the zero-padded [tap][Ci][16] layout of the bf16 mode
    int R, segs;              // output rows per workgroup, workgroups per image
    unsigned x_bytes;
};

namespace {
constexpr unsigned CO4_OOB = 0xfffffff0u;
typedef unsigned co4_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 co4_bf16x8 __attribute__((ext_vector_type(8)));
}

// HALVES = Ci / 16 (2 | 4), TPW = 16-pixel tiles per wave and row = W / 64 (1 | 2 | 4).  XB: the input is a bf16 view -- a lane's 16-byte
// load then holds EIGHT channels (k 8 kq + j of load h' <-> channel 32 h' + 8 kq + j: the filter rows follow the same map): the B operand of a
// bf16 MFMA as it stands; everything after the MFMAs is that of the fp32 form.
// SPLIT (fp32 x, option split6): the same bf16 MFMAs with BOTH operands as three bf16 terms -- a lane's two 16-byte loads of a 32-channel group are its
// eight k-slots (slot 4 u + v <-> channel 32 g + 16 u + 4 kq + v), split in registers (v = hi + mid + lo, each the bf16 rounding of what is left), and the six
// products of order <= 2 are summed in fp32: (hi, hi) (mid, hi) (hi, mid) (lo, hi) (hi, lo) (mid, mid); what is dropped is below 2^-23 of the product.
// 18 bf16 MFMAs of 16 cycles per 32 channels and tile against 24 fp32 ones of 32.
template <int HALVES, int TPW, bool XB = false, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_co4_kernel(const Co4Params p) {
    static_assert(!(XB && SPLIT), "SPLIT is the fp32-input form");
    constexpr int CI = 16 * HALVES, KS = CI / 4;
    constexpr int NLD = XB ? HALVES / 2 : HALVES;      // 16-byte loads per lane and tile
    extern __shared__ __attribute__((aligned(16))) float zs[];      // [9][W + 2][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int W = p.W, WP = W + 2;
    const int b = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = b / p.segs, r0 = (b - n * p.segs) * p.R, r1 = min(r0 + p.R, p.H);

    // A = filter rows: row i = 16 rt + l16 <-> cout l16 >> 2, tap 4 rt + (l16 & 3); k-step 4 h + j <-> channel 16 h + 4 kq + j
    // XB: the GEMM runs on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16: the 32 channels of a 16-byte load per lane are ONE MFMA instead of eight fp32 ones,
    // at 16 instead of 32 cycles each).  The pixel operand is bf16 already; the fp32 filter is split into THREE bf16 terms, w = hi + mid + lo (each the bf16
    // rounding of what the terms before it left: 3 x 8 mantissa bits cover fp32's 24), so the products are those of the fp32 form, exactly, and only the order
    // of the fp32 sum differs -- 9 bf16 MFMAs per 32 channels and tile (144 cycles) against 24 fp32 ones (768).
    constexpr int NG = XB ? NLD : HALVES / 2;          // 32-channel groups
    float a[(XB || SPLIT) ? 1 : KS][3];
    co4_bf16x8 ab[(XB || SPLIT) ? NG : 1][3][3];       // [32-channel group][row tile][term]
    {
        const int co = l16 >> 2, tl = l16 & 3;
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            const int tap = 4 * rt + tl;
            const int tf = p.flip ? 8 - tap : tap;
            if (XB || SPLIT) {
#pragma unroll
                for (int h = 0; h < NG; ++h)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ch = XB ? 32 * h + 8 * kq + j : 32 * h + 16 * (j >> 2) + 4 * kq + (j & 3);
                        const float wv = tap < 9 ? p.w[(tf * CI + ch) * p.wld + co] : 0.f;
                        const __bf16 hi = (__bf16)wv; const float r1 = wv - (float)hi;
                        const __bf16 mid = (__bf16)r1; const __bf16 lo = (__bf16)(r1 - (float)mid);
                        ab[h][rt][0][j] = hi; ab[h][rt][1][j] = mid; ab[h][rt][2][j] = lo;
                    }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int ch = 16 * (ks >> 2) + 4 * kq + (ks & 3);
                    a[ks][rt] = tap < 9 ? p.w[(tf * CI + ch) * p.wld + co] : 0.f;
                }
            }
        }
    }
    for (int i = tid; i < 9 * WP * 4; i += 256) zs[i] = 0.f;       // the halo columns stay zero
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    co4_u32x4 xr[TPW][NLD];
    auto load_tile = [&](int t, int r) {              // tile t of the wave (pixels 16 (wave + 4 t) ..) of input row r; outside -> zeros
        const int px = 16 * (wave + 4 * t) + l16;
        const bool ok = (unsigned)r < (unsigned)p.H && r >= r0 - 1 && r <= r1;
        const unsigned base = (XB ? 2u : 4u) * (unsigned)(((n * p.H + r) * W + px) * p.ldx + (XB ? 8 : 4) * kq);
#pragma unroll
        for (int h = 0; h < NLD; ++h) xr[t][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(ok ? base + 64u * h : CO4_OOB), 0, 0);
    };
    const float4 bias4 = p.bias ? make_float4(p.bias[0], p.bias[1], p.bias[2], p.bias[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 runA = make_float4(0.f, 0.f, 0.f, 0.f), runB = runA;   // output rows r + 1 and r so far, column tid
    auto tap_row = [&](int ty, int pcol) {
        const float4 z0 = *reinterpret_cast<const float4*>(zs + ((3 * ty + 0) * WP + pcol + 0) * 4);
        const float4 z1 = *reinterpret_cast<const float4*>(zs + ((3 * ty + 1) * WP + pcol + 1) * 4);
        const float4 z2 = *reinterpret_cast<const float4*>(zs + ((3 * ty + 2) * WP + pcol + 2) * 4);
        return make_float4((z0.x + z1.x) + z2.x, (z0.y + z1.y) + z2.y, (z0.z + z1.z) + z2.z, (z0.w + z1.w) + z2.w);
    };

#pragma unroll
    for (int t = 0; t < TPW; ++t) load_tile(t, r0 - 1);
    __syncthreads();
    for (int r = r0 - 1; r <= r1; ++r) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if (SPLIT) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const unsigned u[8] = {xr[t][2 * g].x, xr[t][2 * g].y, xr[t][2 * g].z, xr[t][2 * g].w, xr[t][2 * g + 1].x, xr[t][2 * g + 1].y, xr[t][2 * g + 1].z, xr[t][2 * g + 1].w};
                    co4_bf16x8 xh, xm, xl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = __uint_as_float(u[j]);
                        const __bf16 hi = (__bf16)v; const float r1 = v - (float)hi;
                        const __bf16 mid = (__bf16)r1;
                        xh[j] = hi; xm[j] = mid; xl[j] = (__bf16)(r1 - (float)mid);
                    }
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) {   // smallest products first
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[g][rt][1], xm, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[g][rt][0], xl, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[g][rt][2], xh, acc[rt], 0, 0, 0);
                    }
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) {
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[g][rt][0], xm, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[g][rt][1], xh, acc[rt], 0, 0, 0);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[g][rt][0], xh, acc[rt], 0, 0, 0);
                    }
                }
            } else
#pragma unroll
            for (int h = 0; h < NLD; ++h) {
                if (XB) {
                    const co4_bf16x8 xb = __builtin_bit_cast(co4_bf16x8, xr[t][h]);
#pragma unroll
                    for (int term = 2; term >= 0; --term)      // smallest terms first
#pragma unroll
                        for (int rt = 0; rt < 3; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[h][rt][term], xb, acc[rt], 0, 0, 0);
                } else {
                const float xv[4] = {__uint_as_float(xr[t][h].x), __uint_as_float(xr[t][h].y), __uint_as_float(xr[t][h].z), __uint_as_float(xr[t][h].w)};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * h + j][rt], xv[j], acc[rt], 0, 0, 0);
                }
            }
            load_tile(t, r + 1);                      // the registers just multiplied take the next row's pixels
            // Z[tap = 4 rt + reg][pixel][co = kq]: lanes (pixel, co) are 64 consecutive words
            float* zd = zs + (16 * (wave + 4 * t) + l16 + 1) * 4 + kq;
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * rt + q < 9) zd[(4 * rt + q) * WP * 4] = acc[rt][q];
        }
        __syncthreads();
        if (tid < W) {
            // taps (ty, tx) of input row r feed output row r - ty + 1 at column q - tx + 1: column p reads q = p + tx - 1 (+ 1 halo)
            const float4 g0 = tap_row(0, tid), g1 = tap_row(1, tid), g2 = tap_row(2, tid);
            float4 o = make_float4((runB.x + g2.x) + bias4.x, (runB.y + g2.y) + bias4.y, (runB.z + g2.z) + bias4.z, (runB.w + g2.w) + bias4.w);
            if (p.lrelu) { o.x = o.x > 0.f ? o.x : 0.2f * o.x; o.y = o.y > 0.f ? o.y : 0.2f * o.y; o.z = o.z > 0.f ? o.z : 0.2f * o.z; o.w = o.w > 0.f ? o.w : 0.2f * o.w; }
            if (r - 1 >= r0) *reinterpret_cast<float4*>(p.y + ((long long)(n * p.H + r - 1) * W + tid) * p.ldy) = o;
            runB = make_float4(runA.x + g1.x, runA.y + g1.y, runA.z + g1.z, runA.w + g1.w);
            runA = g0;
        }
        __syncthreads();
    }
}

// returns MRDIS_EUNSUPPORTED outside what the kernel covers
int mrdis_run_co4(const void* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co,
                  int flip, int lrelu, hipStream_t s, int x_bf16, int wld) {
    if (Co != 4 || (Ci != 32 && Ci != 64) || (W != 64 && W != 128 && W != 256) || ldx % (x_bf16 ? 8 : 4) != 0 || ldy % 4 != 0) return MRDIS_EUNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0 || mrdis_opt(MRDIS_OPT_NOW16) || (wld != 4 && wld != 16)) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W < 65536) return MRDIS_EUNSUPPORTED;
    const long long xb = (x_bf16 ? 2LL : 4LL) * (((long long)N * H * W - 1) * ldx + Ci);
    if (xb >= 0x7fffffffLL || (long long)N * H * W * ldy >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    Co4Params p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.ldx = ldx; p.ldy = ldy; p.flip = flip; p.lrelu = lrelu;
    p.x_bytes = (unsigned)xb; p.wld = wld;
    int segs = mrdis_cdiv(512, N);                     // ~512 workgroups, each a run of consecutive rows of one image (768 - 2048: no faster)
    if (segs > H / 4) segs = H / 4 > 0 ? H / 4 : 1;
    p.R = mrdis_cdiv(H, segs); p.segs = mrdis_cdiv(H, p.R);
    const size_t lds = sizeof(float) * (size_t)9 * (W + 2) * 4;
    const dim3 grid(N * p.segs), block(256);
    const bool split = !x_bf16 && (mrdis_opt(MRDIS_OPT_SPLIT6) == 1 || mrdis_opt(MRDIS_OPT_SPLIT6) == 3 || mrdis_opt(MRDIS_OPT_SPLIT6) == 4);      // (option split6: mrdis_conv.hip, run_c4conv)
#define CO4_LAUNCH(HV, TP) { if (x_bf16) MRDIS_LAUNCH((conv3x3_co4_kernel<HV, TP, true>), grid, block, lds, s, p); \
                             else if (split) MRDIS_LAUNCH((conv3x3_co4_kernel<HV, TP, false, true>), grid, block, lds, s, p); \
                             else MRDIS_LAUNCH((conv3x3_co4_kernel<HV, TP, false>), grid, block, lds, s, p); }
    if (split) mrdis_count(MRDIS_CNT_SPLIT6_CO4);
    const int tpw = W / 64;
    if (Ci == 64) { if (tpw == 4) CO4_LAUNCH(4, 4) else if (tpw == 2) CO4_LAUNCH(4, 2) else CO4_LAUNCH(4, 1) }
    else { if (tpw == 4) CO4_LAUNCH(2, 4) else if (tpw == 2) CO4_LAUNCH(2, 2) else CO4_LAUNCH(2, 1) }
#undef CO4_LAUNCH
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
