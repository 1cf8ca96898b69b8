// mrdis_conv.hip -- NHWC im2col-free direct convolution for gfx950 (MI355X), fp32.
//
// One "tap-table" formulation covers everything the path needs
// (reference: F.conv2d at src/model.py:2104 and its autograd backward):
//
//   out[n, a*os+oh0, b*os+ow0, co] = bias[co] +
//        sum_t sum_ci in[n, a*is + dh[t], b*is + dw[t], ci] * w[widx[t]][ci][co]
//
//   forward conv          : is = stride, os = 1, dh = r - pad
//   data gradient, s = 1  : is = 1, os = 1, dh = pad - r, w = [T][Co][Ci]
//   data gradient, s = 2  : four output-parity classes, os = 2, oh0 = parity,
//                           each class keeps only the taps that hit it
//
// Kernel shape: a workgroup (256 threads = 4 waves) owns BM = 128 output
// positions (NB images x TH x TW) x BN output channels.  Per KC-channel chunk
// it stages the halo'd input tile ONCE in LDS ([pixel][KC+1], odd stride ->
// conflict-free ds_read_b32 for the MFMA A operand) together with the
// [tap][KC][BN] filter slab, then runs every tap out of LDS:
// v_mfma_f32_32x32x2_f32, A = 32 positions x 2 channels, B = 2 channels x 32
// couts, exact-fp32 accumulate (a k-ordered fmaf chain, so results match an
// fp32 reference to rounding).  blockIdx is remapped so that the tiles an XCD
// works on are contiguous (shared halos / filter slabs hit that XCD's L2).
//
// The weight gradient uses the transposed product: M = (tap, ci) flattened
// into 32-row sub-tiles, N = 32 couts, K = output positions, split over
// workgroups and waves; partial 32x32 slabs are written once and summed in a
// fixed order by a second kernel (bit-reproducible, no float atomics).
#include "mrdis_common.h"
#include "mrdis_tapconv.h"
#include "mrdis_s6conv.h"
#include <stdlib.h>
#include <type_traits>

// 256 B of zeros: the source of LDS-DMA lanes that fall outside the image / channel range
__device__ float g_mrdis_zero_page[64];
#define TC_TAB_INTS 320   // tapconv16: tab_in[128] tab_out[128] tap_xoff[16] tap_widx[16] + pad (tapconv_kernel: 2*BM + 64)

// (bx, gx): workgroup index and count of THIS launch slice -- blockIdx.x / gridDim.x for a plain launch, the class's own range when the
// four parity classes of a stride-2 data gradient share one launch (tapconv_pack_kernel)
// SK = 2 (half-filled 64-position tiles of the small maps, BN = 32): the two position waves that would multiply empty positions take the
// second half of every chunk's channels instead and the pairs add their accumulators through LDS in a fixed order before the epilogue.  A
// wave's tile is ONE accumulator, i.e. a chain of K / 2 dependent 64-cycle MFMAs (128 -> 256 at 8x8: 1152 of them = 35 us whatever the
// staging does); two waves per tile halve the chain.
template <int KC, int BN, int MODE, int BM, int SK = 1>   // staging: 0 generic loops | 1 hoisted descriptors + register prefetch; BM positions per workgroup
__device__ __forceinline__ void tapconv_body(const TapConvParams& p, const int bx, const int gx) {
    constexpr int S = KC + 1;
    constexpr int WAVES_N = (BN == 32) ? 1 : 2;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int MSUB = (BM / 32) / WAVES_M;
    constexpr int NSUB = (BN / 32) / WAVES_N;
    constexpr int KCW = KC / SK;                  // channels of a chunk one wave multiplies
    static_assert(SK == 1 || (SK == 2 && BN == 32 && BM == 128 && MODE == 1 && KC >= 8), "split-K form: 64 positions x 32 couts, two waves per 32 x 32 block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab_in = reinterpret_cast<int*>(smem);
    int* tab_out = tab_in + BM;
    int* tap_xoff = tab_in + 2 * BM;
    int* tap_widx = tab_in + 2 * BM + 16;
    float* ws = smem + (2 * BM + 64);
    float* xs = ws + p.ntaps * KC * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_mr = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int wave_m = (SK == 2) ? (wave_mr & 1) : wave_mr;       // position block of the wave
    const int khalf = (SK == 2) ? (wave_mr >> 1) : 0;             // which half of a chunk's channels

    int bid = mrdis_xcd_remap(bx, gx);
    const int cot = bid % p.coTiles;
    int tile = bid / p.coTiles;
    const int tb = tile % p.tilesB; tile /= p.tilesB;
    const int ta = tile % p.tilesA;
    const int tn = tile / p.tilesA;
    const int a0 = ta * p.TH, b0 = tb * p.TW, n0 = tn * p.NB, co0 = cot * BN;

    for (int m = tid; m < BM; m += 256) {
        const int npos = p.NB * p.TH * p.TW;
        int tin = 0, tout = -1;
        if (m < npos) {
            const int nb = m / (p.TH * p.TW);
            const int rem = m - nb * p.TH * p.TW;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = ((nb * p.TinH + ty * p.is) * p.TinW + tx * p.is) * S;
            const int n = n0 + nb, a = a0 + ty, b = b0 + tx;
            if (n < p.N && a < p.A && b < p.B)
                tout = (n * p.Hout + a * p.os + p.oh0) * p.Wout + b * p.os + p.ow0;
        }
        tab_in[m] = tin;
        tab_out[m] = tout;
    }
    if (tid >= 256 - MRDIS_MAX_TAPS) {
        const int t = tid - (256 - MRDIS_MAX_TAPS);
        if (t < p.ntaps) {
            tap_xoff[t] = ((p.dh[t] - p.dh_min) * p.TinW + (p.dw[t] - p.dw_min)) * S;
            tap_widx[t] = p.widx[t];
        }
    }
    __syncthreads();

    int abase[MSUB];
#pragma unroll
    for (int i = 0; i < MSUB; ++i) abase[i] = tab_in[(wave_m * MSUB + i) * 32 + (lane & 31)] + (lane >> 5) + khalf * KCW;
    const int bbase = ((lane >> 5) + khalf * KCW) * BN + wave_n * NSUB * 32 + (lane & 31);

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int i = 0; i < MSUB; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int tinHW = p.TinH * p.TinW;
    const int npix_in = p.NB * tinHW;
    const int h_org = a0 * p.is + p.dh_min, w_org = b0 * p.is + p.dw_min;

    // ---- one KC-chunk of MFMAs, every tap out of LDS.  Operands of the NEXT group of k-steps are read
    // (register double buffer) before the MFMAs of the current group issue, and sched_barriers keep
    // that order, so a 64-cycle MFMA never waits on an LDS round trip; the next tap's offset is
    // fetched a whole tap ahead.
    constexpr int MPS = MSUB * NSUB;                         // MFMAs per k-step
    constexpr int G0 = MPS >= 4 ? 1 : (MPS == 2 ? 2 : 4);
    constexpr int G = (KCW / 2 < G0) ? KCW / 2 : G0;         // k-steps per group (>= 256 MFMA cycles when possible)
    constexpr int NG = (KCW / 2) / G;
    auto compute_chunk = [&]() {
        float an[G][MSUB], bn[G][NSUB];
        int toff = tap_xoff[0];
        int toff_n = tap_xoff[p.ntaps > 1 ? 1 : 0];
#pragma unroll
        for (int s_ = 0; s_ < G; ++s_) {
#pragma unroll
            for (int i = 0; i < MSUB; ++i) an[s_][i] = xs[abase[i] + toff + 2 * s_];
#pragma unroll
            for (int j = 0; j < NSUB; ++j) bn[s_][j] = ws[bbase + 2 * s_ * BN + j * 32];
        }
        for (int t = 0; t < p.ntaps; ++t) {
            const float* wt = ws + t * (KC * BN) + bbase;
            const int tn = (t + 1 < p.ntaps) ? t + 1 : t;
            const float* wtn = ws + tn * (KC * BN) + bbase;
            const int toff_nn = tap_xoff[(t + 2 < p.ntaps) ? t + 2 : tn];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float av[G][MSUB], bv[G][NSUB];
#pragma unroll
                for (int s_ = 0; s_ < G; ++s_) {
#pragma unroll
                    for (int i = 0; i < MSUB; ++i) av[s_][i] = an[s_][i];
#pragma unroll
                    for (int j = 0; j < NSUB; ++j) bv[s_][j] = bn[s_][j];
                }
#pragma unroll
                for (int s_ = 0; s_ < G; ++s_) {
                    if (g + 1 < NG) {
                        const int kk = (g + 1) * G + s_;
#pragma unroll
                        for (int i = 0; i < MSUB; ++i) an[s_][i] = xs[abase[i] + toff + 2 * kk];
#pragma unroll
                        for (int j = 0; j < NSUB; ++j) bn[s_][j] = wt[2 * kk * BN + j * 32];
                    } else {
#pragma unroll
                        for (int i = 0; i < MSUB; ++i) an[s_][i] = xs[abase[i] + toff_n + 2 * s_];
#pragma unroll
                        for (int j = 0; j < NSUB; ++j) bn[s_][j] = wtn[2 * s_ * BN + j * 32];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s_ = 0; s_ < G; ++s_)
#pragma unroll
                    for (int i = 0; i < MSUB; ++i)
#pragma unroll
                        for (int j = 0; j < NSUB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[s_][j], av[s_][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            toff = toff_n; toff_n = toff_nn;
        }
    };


    if constexpr (MODE != 0) {
        // ---- hoisted staging: what each thread fetches (tile geometry, filter rows) is chunk-invariant, so the
        // divisions and 64-bit address arithmetic run once per workgroup and a chunk costs one add per item
        // (the generic loops below spend ~3 VALU instructions per MFMA on it); the global loads of chunk c+1 are
        // in flight (registers) while chunk c runs out of LDS.  Measured against the generic loops: 7-16 %
        // faster on every layer shape of the step (tools/ab_lib.py, MRDIS_DEBUG_MODE=0).
        constexpr int XR = 6, WR = 9, QX = KC / 4, QW = BN / 4;
        const int nx = npix_in * QX, nw = p.ntaps * KC * QW;
        int xg[XR], xl[XR], wg[WR];
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + it * 256;
            xg[it] = -1; xl[it] = -1;
            if (idx < nx) {
                const int pi = idx / QX, q = idx - pi * QX;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix;
                xl[it] = pi * S + 4 * q;
                if (n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win)
                    xg[it] = ((n * p.Hin + h) * p.Win + w_) * p.ldin + 4 * q;       // host: < 2^31 elements
            }
        }
#pragma unroll
        for (int it = 0; it < WR; ++it) {
            const int idx = tid + it * 256;
            wg[it] = -1;
            if (idx < nw) {
                const int row = idx / QW, q = idx - row * QW;
                const int t = row / KC, k = row - t * KC;
                const int co = co0 + 4 * q;
                if (co < p.Cout) wg[it] = (tap_widx[t] * p.Cin + k) * p.Cout + co;
            }
        }
        float4 xr[XR], wr[WR];
        auto load_chunk = [&](int c0) {
#pragma unroll
            for (int it = 0; it < XR; ++it) {
                const int q = (tid + it * 256) % QX;
                xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xg[it] >= 0 && c0 + 4 * q < p.Cin) xr[it] = *reinterpret_cast<const float4*>(p.in + xg[it] + c0);
            }
            const float* wc = p.w + (long long)c0 * p.Cout;
#pragma unroll
            for (int it = 0; it < WR; ++it) {
                const int k = ((tid + it * 256) / QW) % KC;
                wr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wg[it] >= 0 && c0 + k < p.Cin) wr[it] = *reinterpret_cast<const float4*>(wc + wg[it]);
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int it = 0; it < XR; ++it)
                if (xl[it] >= 0) { float* d = xs + xl[it]; d[0] = xr[it].x; d[1] = xr[it].y; d[2] = xr[it].z; d[3] = xr[it].w; }
#pragma unroll
            for (int it = 0; it < WR; ++it) {
                const int idx = tid + it * 256;
                if (idx < nw) *reinterpret_cast<float4*>(ws + 4 * idx) = wr[it];
            }
        };
        load_chunk(0);
        store_chunk();
        __syncthreads();
        for (int c0 = 0; c0 < p.Cin; c0 += KC) {
            const bool more = c0 + KC < p.Cin;
            if (more) load_chunk(c0 + KC);
            compute_chunk();
            if (more) { __syncthreads(); store_chunk(); __syncthreads(); }
        }
    } else
    for (int c0 = 0; c0 < p.Cin; c0 += KC) {
        if (c0) __syncthreads();
        // ---- stage the halo'd input tile: xs[pixel][KC+1]
        if (p.vec_in) {
            constexpr int Q = KC / 4;
            for (int idx = tid; idx < npix_in * Q; idx += 256) {
                const int pi = idx / Q, q = idx - pi * Q;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix, c = c0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win && c < p.Cin)
                    v = *reinterpret_cast<const float4*>(p.in + ((long long)(n * p.Hin + h) * p.Win + w_) * p.ldin + c);
                float* d = xs + pi * S + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < npix_in * KC; idx += 256) {
                const int pi = idx / KC, k = idx - pi * KC;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix, c = c0 + k;
                float v = 0.f;
                if (n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win && c < p.Cin)
                    v = p.in[((long long)(n * p.Hin + h) * p.Win + w_) * p.ldin + c];
                xs[pi * S + k] = v;
            }
        }
        // ---- stage the filter slab: ws[tap][KC][BN]
        if (p.vec_w) {
            constexpr int Q = BN / 4;
            const int total = p.ntaps * KC * Q;
            for (int idx = tid; idx < total; idx += 256) {
                const int row = idx / Q, q = idx - row * Q;
                const int t = row / KC, k = row - t * KC;
                const int c = c0 + k, co = co0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < p.Cin && co < p.Cout)
                    v = *reinterpret_cast<const float4*>(p.w + ((long long)tap_widx[t] * p.Cin + c) * p.Cout + co);
                *reinterpret_cast<float4*>(ws + row * BN + 4 * q) = v;
            }
        } else {
            const int total = p.ntaps * KC * BN;
            for (int idx = tid; idx < total; idx += 256) {
                const int row = idx / BN, j = idx - row * BN;
                const int t = row / KC, k = row - t * KC;
                const int c = c0 + k, co = co0 + j;
                float v = 0.f;
                if (c < p.Cin && co < p.Cout) v = p.w[((long long)tap_widx[t] * p.Cin + c) * p.Cout + co];
                ws[idx] = v;
            }
        }
        __syncthreads();
        compute_chunk();
    }

    // ---- epilogue.  The MFMA operands are swapped (A = filter, B = pixels), so D is [cout][position]:
    // a lane owns ONE output position per M sub-tile (column lane&31) and the couts
    // (r&3) + 8*(r>>2) + 4*(lane>>5) -> groups of four consecutive couts = one 16-byte store.  One address
    // per lane and sub-tile, 4 stores per accumulator instead of 16 (the epilogue is issue-bound).
    if constexpr (SK == 2) {
        static_assert(MSUB == 1 && NSUB == 1, "one accumulator per wave");
        __syncthreads();                              // the operand images are dead: the filter slab's first 8 KB carry the partials
        float* red = ws;
        if (khalf == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave_m * 16 + r) * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (khalf == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += red[(wave_m * 16 + r) * 64 + lane];
    }
    const bool lrelu = (p.epilogue & MRDIS_EPI_LRELU) != 0;
    const int half = lane >> 5;
    const bool vec_out = (p.ldout % 4 == 0) && (((uintptr_t)p.out & 15) == 0);
    const bool vec_bias = p.bias != nullptr && (((uintptr_t)p.bias & 15) == 0);
#pragma unroll
    for (int i = 0; i < MSUB; ++i) {
        const int po = tab_out[(wave_m * MSUB + i) * 32 + (lane & 31)];
        if (po < 0) continue;
        float* dst = p.out + (long long)po * p.ldout;
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + (wave_n * NSUB + j) * 32 + 8 * g + 4 * half;
                if (co >= p.Cout) continue;
                float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                const bool full = co + 3 < p.Cout;
                if (p.bias != nullptr) {
                    if (vec_bias && full) {
                        const float4 bq = *reinterpret_cast<const float4*>(p.bias + co);
                        v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
                    } else {
                        v.x += p.bias[co];
                        if (co + 1 < p.Cout) v.y += p.bias[co + 1];
                        if (co + 2 < p.Cout) v.z += p.bias[co + 2];
                        if (co + 3 < p.Cout) v.w += p.bias[co + 3];
                    }
                }
                if (lrelu) {
                    v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y;
                    v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w;
                }
                if (vec_out && full) *reinterpret_cast<float4*>(dst + co) = v;
                else {
                    dst[co] = v.x;
                    if (co + 1 < p.Cout) dst[co + 1] = v.y;
                    if (co + 2 < p.Cout) dst[co + 2] = v.z;
                    if (co + 3 < p.Cout) dst[co + 3] = v.w;
                }
            }
        }
    }
}

template <int KC, int BN, int MODE, int BM, int SK = 1>
__global__ __launch_bounds__(256) void tapconv_kernel(const TapConvParams p) {
    tapconv_body<KC, BN, MODE, BM, SK>(p, (int)blockIdx.x, (int)gridDim.x);
}
// The four output-parity classes of a stride-2 data gradient (each its own tap set and output grid, disjoint output pixels) in ONE
// launch: blockIdx.y = class.  On the small maps of the encoders a class alone fills a quarter of the chip (128 workgroups at
// 256 -> 256 on 16x16) and the four launches ran back to back.
struct TapConvPack { TapConvParams c[4]; int nblk[4]; };
template <int KC, int BN, int MODE, int BM>
__global__ __launch_bounds__(256) void tapconv_pack_kernel(const TapConvPack pk) {
    const int y = blockIdx.y;
    if ((int)blockIdx.x >= pk.nblk[y]) return;
    tapconv_body<KC, BN, MODE, BM>(pk.c[y], (int)blockIdx.x, pk.nblk[y]);
}

// ---------------------------------------------------------------- narrow-output variant (Cout <= 16)
// Layers with 4 / 7 / 16 output channels (ana_dec.output, the data gradients of every `si_layers`,
// sp6.out, the 1x1 decoder head, first-layer data gradients) waste 2-8x of a 32-wide cout tile.  This
// variant uses v_mfma_f32_16x16x4_f32 (A = filter [16 couts x 4 ch], B = pixels [4 ch x 16 positions],
// same FLOP rate, 32-cycle issue): workgroup = 128 positions x 16 couts, a wave owns two 16-position
// sub-tiles that share the A operand.  D[cout][position]: lane = position, registers = 4 consecutive
// couts -> one 16-byte store per sub-tile.
// THIN4 (Cout <= 4: ana_dec.output, the data gradients of every si_layers): even a 16-wide MFMA tile is 75 % padding
// there, so the product runs as packed FMAs instead -- thread = (position, channel half of the chunk), the pixel's
// channels come out of the same padded LDS tile (conflict-free across positions), the 4 filter values of a
// (tap, channel) are one wave-uniform 16-byte LDS read.
template <int KC, bool THIN4>
__global__ __launch_bounds__(256) void tapconv16_kernel(const TapConvParams p) {
    constexpr int S = THIN4 ? KC + 4 : KC + 1, BN = 16;      // THIN4 reads pixels as 16-byte quads: pitch 20 floats (aligned, conflict-free)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab_in = reinterpret_cast<int*>(smem);
    int* tab_out = tab_in + 128;
    int* tap_xoff = tab_in + 256;
    int* tap_widx = tab_in + 272;
    float* ws = smem + TC_TAB_INTS;                  // [tap][KC][16]
    float* xs = ws + p.ntaps * KC * BN;              // [pixel][KC+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;

    int bid = mrdis_xcd_remap(blockIdx.x, gridDim.x);
    int tile = bid;
    const int tb = tile % p.tilesB; tile /= p.tilesB;
    const int ta = tile % p.tilesA;
    const int tn = tile / p.tilesA;
    const int a0 = ta * p.TH, b0 = tb * p.TW, n0 = tn * p.NB;

    if (tid < 128) {
        const int m = tid, npos = p.NB * p.TH * p.TW;
        int tin = 0, tout = -1;
        if (m < npos) {
            const int nb = m / (p.TH * p.TW);
            const int rem = m - nb * p.TH * p.TW;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = ((nb * p.TinH + ty * p.is) * p.TinW + tx * p.is) * S;
            const int n = n0 + nb, a = a0 + ty, b = b0 + tx;
            if (n < p.N && a < p.A && b < p.B) tout = (n * p.Hout + a * p.os + p.oh0) * p.Wout + b * p.os + p.ow0;
        }
        tab_in[m] = tin; tab_out[m] = tout;
    } else if (tid < 128 + MRDIS_MAX_TAPS) {
        const int t = tid - 128;
        if (t < p.ntaps) {
            tap_xoff[t] = ((p.dh[t] - p.dh_min) * p.TinW + (p.dw[t] - p.dw_min)) * S;
            tap_widx[t] = p.widx[t];
        }
    }
    __syncthreads();
    int bbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bbase[i] = tab_in[(wave * 2 + i) * 16 + l16] + kq;
    const int abase = kq * BN + l16;

    f32x4 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tinHW = p.TinH * p.TinW, npix_in = p.NB * tinHW;
    const int h_org = a0 * p.is + p.dh_min, w_org = b0 * p.is + p.dw_min;

    // THIN4 thread roles: positions (tid & 63) and (tid & 63) + 64, channel quarter = wave (4 channels of the 16-channel chunk):
    // one 16-byte filter read (wave-uniform) now feeds two positions and a pixel's four channels are one 16-byte read --
    // 54 LDS instructions per chunk and thread instead of 144 (the kernel was LDS-issue bound)
    const int tq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tb0 = tab_in[tid & 63] + 4 * tq, tb1 = tab_in[(tid & 63) + 64] + 4 * tq;
    float acc4[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // (reading the filter values as scalar loads from global memory instead of the LDS broadcast was tried: slower, the
    // scalar and LDS counters share lgkmcnt and the waits serialise)
    auto compute_chunk = [&](int) {
        if constexpr (THIN4) {
            for (int t = 0; t < p.ntaps; ++t) {
                const int toff = tap_xoff[t];
                const float4 x0 = *reinterpret_cast<const float4*>(xs + tb0 + toff);
                const float4 x1 = *reinterpret_cast<const float4*>(xs + tb1 + toff);
                const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
                const float* wr = ws + (t * KC + 4 * tq) * BN;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wr + kk * BN);
                    acc4[0][0] += xa[kk] * w4.x; acc4[0][1] += xa[kk] * w4.y; acc4[0][2] += xa[kk] * w4.z; acc4[0][3] += xa[kk] * w4.w;
                    acc4[1][0] += xb[kk] * w4.x; acc4[1][1] += xb[kk] * w4.y; acc4[1][2] += xb[kk] * w4.z; acc4[1][3] += xb[kk] * w4.w;
                }
            }
        } else {
            for (int t = 0; t < p.ntaps; ++t) {
                const int toff = tap_xoff[t];
                const float* wt = ws + t * (KC * BN) + abase;
#pragma unroll
                for (int q = 0; q < KC / 4; ++q) {
                    const float av = wt[4 * q * BN];
                    const float b0v = xs[bbase[0] + toff + 4 * q], b1v = xs[bbase[1] + toff + 4 * q];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0v, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1v, acc[1], 0, 0, 0);
                }
            }
        }
    };
    if (p.prefetch) {
        // hoisted staging descriptors + register prefetch of the next chunk (as tapconv_kernel MODE 1); the filter
        // slab is fetched as 16-byte pieces (Cout % 4 == 0) and scattered into its [tap][KC][16] rows
        constexpr int XR = 6, WR = 3, QX = KC / 4;
        const int nx = npix_in * QX, cq = p.Cout >> 2, nw = p.ntaps * KC * cq;
        int xg[XR], xl[XR], wg[WR], wl[WR], wk[WR];
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + it * 256;
            xg[it] = -1; xl[it] = -1;
            if (idx < nx) {
                const int pi = idx / QX, q = idx - pi * QX;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix;
                xl[it] = pi * S + 4 * q;
                if (n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win)
                    xg[it] = ((n * p.Hin + h) * p.Win + w_) * p.ldin + 4 * q;
            }
        }
#pragma unroll
        for (int it = 0; it < WR; ++it) {
            const int idx = tid + it * 256;
            wg[it] = -1; wl[it] = 0; wk[it] = 0;
            if (idx < nw) {
                const int row = idx / cq, q = idx - row * cq;
                const int t = row / KC, k = row - t * KC;
                wg[it] = (tap_widx[t] * p.Cin + k) * p.Cout + 4 * q;
                wl[it] = row * BN + 4 * q; wk[it] = k;
            }
        }
        float4 xr[XR], wr[WR];
        auto load_chunk = [&](int c0) {
#pragma unroll
            for (int it = 0; it < XR; ++it) {
                const int q = (tid + it * 256) % QX;
                xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xg[it] >= 0 && c0 + 4 * q < p.Cin) xr[it] = *reinterpret_cast<const float4*>(p.in + xg[it] + c0);
            }
            const float* wc = p.w + (long long)c0 * p.Cout;
#pragma unroll
            for (int it = 0; it < WR; ++it) {
                wr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wg[it] >= 0 && c0 + wk[it] < p.Cin) wr[it] = *reinterpret_cast<const float4*>(wc + wg[it]);
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int it = 0; it < XR; ++it)
                if (xl[it] >= 0) {
                    float* d = xs + xl[it];
                    // THIN4: pixel pitch 20 floats -> the quad is 16-byte aligned: one ds_write_b128 (four single stores at that
                    // pitch collide up to 4-way: 20 pi + 4 q mod 32; PMC 1.5 conflict cycles per LDS instruction)
                    if constexpr (THIN4) *reinterpret_cast<float4*>(d) = xr[it];
                    else { d[0] = xr[it].x; d[1] = xr[it].y; d[2] = xr[it].z; d[3] = xr[it].w; }
                }
#pragma unroll
            for (int it = 0; it < WR; ++it)
                if (wg[it] >= 0) *reinterpret_cast<float4*>(ws + wl[it]) = wr[it];
        };
        // columns >= Cout of the filter slab stay zero for the whole kernel
        for (int idx = tid; idx < p.ntaps * KC * BN; idx += 256) ws[idx] = 0.f;
        __syncthreads();
        load_chunk(0);
        store_chunk();
        __syncthreads();
        for (int c0 = 0; c0 < p.Cin; c0 += KC) {
            const bool more = c0 + KC < p.Cin;
            if (more) load_chunk(c0 + KC);
            compute_chunk(c0);
            if (more) { __syncthreads(); store_chunk(); __syncthreads(); }
        }
    } else
    for (int c0 = 0; c0 < p.Cin; c0 += KC) {
        if (c0) __syncthreads();
        if (p.vec_in) {
            constexpr int Q = KC / 4;
            for (int idx = tid; idx < npix_in * Q; idx += 256) {
                const int pi = idx / Q, q = idx - pi * Q;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix, c = c0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win && c < p.Cin)
                    v = *reinterpret_cast<const float4*>(p.in + ((long long)(n * p.Hin + h) * p.Win + w_) * p.ldin + c);
                float* d = xs + pi * S + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < npix_in * KC; idx += 256) {
                const int pi = idx / KC, k = idx - pi * KC;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix, c = c0 + k;
                float v = 0.f;
                if (n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win && c < p.Cin)
                    v = p.in[((long long)(n * p.Hin + h) * p.Win + w_) * p.ldin + c];
                xs[pi * S + k] = v;
            }
        }
        {
            const int total = p.ntaps * KC * BN;
            for (int idx = tid; idx < total; idx += 256) {
                const int row = idx >> 4, j = idx & 15;
                const int t = row / KC, k = row - t * KC;
                const int c = c0 + k;
                float v = 0.f;
                if (c < p.Cin && j < p.Cout) v = p.w[((long long)tap_widx[t] * p.Cin + c) * p.Cout + j];
                ws[idx] = v;
            }
        }
        __syncthreads();
        compute_chunk(c0);
    }
    if constexpr (THIN4) {
        // the four channel quarters of a position meet in LDS (fixed order: q0 + q1 + q2 + q3)
        __syncthreads();
        float* red = xs;                                  // [4 quarters][128 positions][4]
#pragma unroll
        for (int h = 0; h < 2; ++h)
            *reinterpret_cast<float4*>(red + ((tq * 128 + (tid & 63) + 64 * h) * 4)) = make_float4(acc4[h][0], acc4[h][1], acc4[h][2], acc4[h][3]);
        __syncthreads();
        if (tid < 128) {
            const int tpos = tid;
            const int po = tab_out[tpos];
            if (po >= 0) {
                const bool lrelu_ = (p.epilogue & MRDIS_EPI_LRELU) != 0;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = ((red[tpos * 4 + r] + red[(128 + tpos) * 4 + r]) + red[(256 + tpos) * 4 + r]) + red[(384 + tpos) * 4 + r];
                    if (p.bias != nullptr && r < p.Cout) v[r] += p.bias[r];
                    if (lrelu_) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
                }
                float* dst = p.out + (long long)po * p.ldout;
                if (p.Cout == 4 && (p.ldout % 4 == 0) && (((uintptr_t)p.out & 15) == 0)) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (r < p.Cout) dst[r] = v[r];
                }
            }
        }
        return;
    }
    // D: col = lane&15 (position), row = (lane>>4)*4 + reg (cout)
    const bool lrelu = (p.epilogue & MRDIS_EPI_LRELU) != 0;
    const bool vec_out = (p.ldout % 4 == 0) && (((uintptr_t)p.out & 15) == 0);
    const int co = 4 * kq;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int po = tab_out[(wave * 2 + i) * 16 + l16];
        if (po < 0 || co >= p.Cout) continue;
        float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (p.bias != nullptr && co + r < p.Cout) v[r] += p.bias[co + r];
            if (lrelu) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
        }
        float* dst = p.out + (long long)po * p.ldout + co;
        if (vec_out && co + 3 < p.Cout) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (co + r < p.Cout) dst[r] = v[r];
        }
    }
}

// ------------------------------------------------------------------ host side
template <int KC, int BN>
static int launch_tapconv_t(const TapConvParams& p, size_t lds, int nblk, int BM, hipStream_t s, bool split_k = false) {
    if constexpr (BN == 32 && KC >= 16) {
        if (split_k && BM == 128 && p.prefetch == 1) {
            MRDIS_LAUNCH((tapconv_kernel<KC, BN, 1, 128, 2>), dim3(nblk), dim3(256), lds, s, p);
            MRDIS_CHECK_LAUNCH();
            return MRDIS_OK;
        }
    }
    if (BM == 256) MRDIS_LAUNCH((tapconv_kernel<KC, BN, 1, 256>), dim3(nblk), dim3(256), lds, s, p);      // BM 256 only with MODE 1
    else if (p.prefetch == 1) MRDIS_LAUNCH((tapconv_kernel<KC, BN, 1, 128>), dim3(nblk), dim3(256), lds, s, p);
    else MRDIS_LAUNCH((tapconv_kernel<KC, BN, 0, 128>), dim3(nblk), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// a planned (not yet launched) tapconv_kernel launch: run_tapconv fills it instead of launching when asked to
struct TapLaunch { TapConvParams p; int KC, BN, BM, nblk; size_t lds; bool set; BConvLaunch b; S6ConvLaunch s6; };
template <int KC, int BN>
static int launch_tapconv_pack_t(const TapLaunch (&L)[4], hipStream_t s) {
    TapConvPack pk;
    int gx = 0; size_t lds = 0;
    for (int k = 0; k < 4; ++k) { pk.c[k] = L[k].p; pk.nblk[k] = L[k].nblk; if (L[k].nblk > gx) gx = L[k].nblk; if (L[k].lds > lds) lds = L[k].lds; }
    const int BM = L[0].BM, pf = L[0].p.prefetch;
    if (BM == 256) MRDIS_LAUNCH((tapconv_pack_kernel<KC, BN, 1, 256>), dim3(gx, 4), dim3(256), lds, s, pk);
    else if (pf == 1) MRDIS_LAUNCH((tapconv_pack_kernel<KC, BN, 1, 128>), dim3(gx, 4), dim3(256), lds, s, pk);
    else MRDIS_LAUNCH((tapconv_pack_kernel<KC, BN, 0, 128>), dim3(gx, 4), dim3(256), lds, s, pk);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

static size_t tapconv_lds(const TapConvParams& p, int KC, int BN, int BM = TC_BM) {
    return sizeof(float) * ((size_t)(2 * BM + 64) + (size_t)p.ntaps * KC * BN + (size_t)p.NB * p.TinH * p.TinW * (KC + 1));
}

// fills tiling fields and launches.  `p` must have geometry + taps set.
static int run_tapconv(TapConvParams p, hipStream_t s, TapLaunch* defer = nullptr) {
    if (defer) { defer->set = false; defer->b.set = false; defer->s6.set = false; }
    if (p.ntaps < 1 || p.ntaps > MRDIS_MAX_TAPS) return MRDIS_EUNSUPPORTED;
    if (p.A <= 0 || p.B <= 0 || p.N <= 0) return MRDIS_OK;   // empty launch
    int dh_max = p.dh[0], dw_max = p.dw[0];
    p.dh_min = p.dh[0]; p.dw_min = p.dw[0];
    for (int t = 1; t < p.ntaps; ++t) {
        if (p.dh[t] < p.dh_min) p.dh_min = p.dh[t];
        if (p.dh[t] > dh_max) dh_max = p.dh[t];
        if (p.dw[t] < p.dw_min) p.dw_min = p.dw[t];
        if (p.dw[t] > dw_max) dw_max = p.dw[t];
    }
    if ((p.dtype == MRDIS_DT_F32_BF16M || p.dtype == MRDIS_DT_BF16) && p.w_bf16) {      // bf16 MFMA operands (mrdis_bf16.hip) where the geometry allows
        const int rc = mrdis_run_bconv(p, dh_max, dw_max, s, defer ? &defer->b : nullptr);
        if (rc != MRDIS_EUNSUPPORTED || p.dtype == MRDIS_DT_BF16) return rc;         // bf16 views never reach the fp32 kernels
    }
    if (p.dtype == MRDIS_DT_BF16) return MRDIS_EUNSUPPORTED;
    if (p.dtype == MRDIS_DT_F32 && p.s6_img) {     // fp32 operands as three bf16 terms on the bf16 matrix pipe (option split6; mrdis_s6conv.hip): the caller brought the filter image
        const int rc = mrdis_run_s6conv(p, p.s6_img, p.s6_taps, dh_max, dw_max, s, defer ? &defer->s6 : nullptr);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    TileChoice tc = choose_tile(p.N, p.A, p.B);
    bool small_map = false;
    // small maps with many channels (the 16x16 / 8x8 levels of the encoders: 2048 output positions, K = 16 taps x 256 channels): 128-position
    // tiles give 16 x (Cout / 32) workgroups, a fraction of the chip, each with a long serial K loop.  Half-filled tiles (64 positions of
    // the 128 a workgroup can hold) double the workgroup count; the matrix pipe is far from busy there, the extra MFMA work is free
    // (round 3: the idle half of the waves takes half of each chunk's channels, tapconv_body SK = 2.  Widening the rule to launches of
    //  256 workgroups -- the 16x16 level -- lost: 128 -> 128 37 -> 57 us, 256 -> 128 66 -> 103 us; those are staging-bound, not chain-bound)
    if (p.Cout > 16 && (long long)mrdis_cdiv(p.A, tc.TH) * mrdis_cdiv(p.B, tc.TW) * mrdis_cdiv(p.N, tc.NB) * mrdis_cdiv(p.Cout, 32) < 200 &&
        (long long)p.N * p.A * p.B >= 256 && p.Cin >= 64 && p.os == 1 && !mrdis_opt(MRDIS_OPT_NOW16))      // (os == 2: the parity classes of a stride-2
                                                                                                    // data gradient already share one launch)
        { tc = choose_tile(p.N, p.A, p.B, 64); small_map = true; }
    p.NB = tc.NB; p.TH = tc.TH; p.TW = tc.TW;
    p.TinH = (p.TH - 1) * p.is + (dh_max - p.dh_min) + 1;
    p.TinW = (p.TW - 1) * p.is + (dw_max - p.dw_min) + 1;
    p.tilesA = mrdis_cdiv(p.A, p.TH); p.tilesB = mrdis_cdiv(p.B, p.TW); p.tilesN = mrdis_cdiv(p.N, p.NB);
    const long long ptiles = (long long)p.tilesA * p.tilesB * p.tilesN;
    if (p.Cout <= 16 && !mrdis_opt(MRDIS_OPT_NO16)) {
        p.vec_in = (p.Cin % 4 == 0) && (p.ldin % 4 == 0) && (((uintptr_t)p.in & 15) == 0);
        p.vec_w = 0; p.prefetch = 0; p.coTiles = 1;
        int KC = p.Cin <= 4 ? 4 : (p.Cin <= 8 ? 8 : 16);
        const bool thin4_want = p.Cout <= 4 && !mrdis_opt(MRDIS_OPT_NOTHIN);
        auto lds16 = [&](int kc) {
            const bool t4 = thin4_want && kc == 16;
            size_t xs_floats = (size_t)p.NB * p.TinH * p.TinW * (kc + (t4 ? 4 : 1));
            if (t4 && xs_floats < 4 * 128 * 4) xs_floats = 4 * 128 * 4;          // the quarter-sum buffer of the THIN4 epilogue reuses the tile
            return sizeof(float) * ((size_t)TC_TAB_INTS + (size_t)p.ntaps * kc * 16 + xs_floats);
        };
        while (lds16(KC) > 64 * 1024 && KC > 4) KC >>= 1;
        if (lds16(KC) > 64 * 1024) return MRDIS_EUNSUPPORTED;
        // hoisted descriptors + register prefetch: vector paths on both operands, item counts within the register arrays,
        // 32-bit element offsets
        {
            const long long npix = (long long)p.NB * p.TinH * p.TinW;
            const bool vec_w16 = (p.Cout % 4 == 0) && (((uintptr_t)p.w & 15) == 0);
            const bool small = (long long)p.N * p.Hin * p.Win * p.ldin < 0x7fffffffLL && (long long)MRDIS_MAX_TAPS * p.Cin * p.Cout < 0x7fffffffLL;
            p.prefetch = (p.vec_in && vec_w16 && small && npix * (KC / 4) <= 6 * 256 && (long long)p.ntaps * KC * (p.Cout / 4) <= 3 * 256) ? 1 : 0;
            if (mrdis_opt(MRDIS_OPT_MODE) >= 0 && p.prefetch) p.prefetch = mrdis_opt(MRDIS_OPT_MODE) ? 1 : 0;
        }
        if (ptiles > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
        const bool thin4 = p.Cout <= 4 && KC == 16 && !mrdis_opt(MRDIS_OPT_NOTHIN);
        if (thin4) MRDIS_LAUNCH((tapconv16_kernel<16, true>), dim3((int)ptiles), dim3(256), lds16(16), s, p);
        else if (KC == 16) MRDIS_LAUNCH((tapconv16_kernel<16, false>), dim3((int)ptiles), dim3(256), lds16(16), s, p);
        else if (KC == 8) MRDIS_LAUNCH((tapconv16_kernel<8, false>), dim3((int)ptiles), dim3(256), lds16(8), s, p);
        else MRDIS_LAUNCH((tapconv16_kernel<4, false>), dim3((int)ptiles), dim3(256), lds16(4), s, p);
        MRDIS_CHECK_LAUNCH();
        return MRDIS_OK;
    }
    // measured policy (tools/sweep.py, B=32 256x256 layer zoo): 64-wide cout tiles beat 128-wide ones on every
    // layer (register pressure halves the residency of the 128 variant); small grids prefer 32.
    int BN = p.Cout <= 32 ? 32 : 64;
    if (BN == 64 && ptiles * mrdis_cdiv(p.Cout, 64) < 256) BN = 32;
    { const int v = (int)mrdis_opt(MRDIS_OPT_BN); if (v == 32 || v == 64) BN = v; }
    p.vec_in = (p.Cin % 4 == 0) && (p.ldin % 4 == 0) && (((uintptr_t)p.in & 15) == 0);
    p.vec_w = (p.Cout % 4 == 0) && (((uintptr_t)p.w & 15) == 0);
    int KC = p.Cin <= 4 ? 4 : (p.Cin <= 8 ? 8 : 16);
    const size_t LDS_MAX = 64 * 1024;
    const long long npix_in = (long long)p.NB * p.TinH * p.TinW;
    auto fits_pf = [&](int kc, int bn) { return npix_in * (kc / 4) <= 6 * 256 && (long long)p.ntaps * kc * (bn / 4) <= 9 * 256; };
    const bool want_pf = p.vec_in && p.vec_w;
    // small maps: a workgroup's time is its chain of chunks (two barriers + one exposed load round trip each, ~0.4 us per 16 channels
    // whatever the tile holds) -- 32-channel chunks halve the chain where the staging registers and LDS still fit (3x3 taps, 32 couts)
    if (small_map && BN == 32 && p.Cin % 32 == 0 && want_pf && fits_pf(32, 32) && tapconv_lds(p, 32, 32) <= LDS_MAX) KC = 32;
    { const int v = (int)mrdis_opt(MRDIS_OPT_KC); if ((v == 4 || v == 8 || v == 16) && v < KC) KC = v; }
    while ((tapconv_lds(p, KC, BN) > LDS_MAX || (want_pf && !fits_pf(KC, BN))) && KC > 4) KC >>= 1;
    while (tapconv_lds(p, KC, BN) > LDS_MAX && BN > 32) BN >>= 1;
    if (tapconv_lds(p, KC, BN) > LDS_MAX) return MRDIS_EUNSUPPORTED;
    // 32-bit element offsets in the hoisted descriptors
    const bool small = (long long)p.N * p.Hin * p.Win * p.ldin < 0x7fffffffLL && (long long)MRDIS_MAX_TAPS * p.Cin * p.Cout < 0x7fffffffLL;
    p.prefetch = (want_pf && fits_pf(KC, BN) && small) ? 1 : 0;
    if (mrdis_opt(MRDIS_OPT_MODE) >= 0 && p.prefetch) p.prefetch = mrdis_opt(MRDIS_OPT_MODE) ? 1 : 0;
    p.coTiles = mrdis_cdiv(p.Cout, BN);
    long long nblk = ptiles * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    size_t lds = tapconv_lds(p, KC, BN);
    int BM = 128;
    // 256-position workgroups: twice the MFMA work per staged filter slab.  Pays on the 32-cout layers (8-10 %, a
    // 32-wide tile reuses each staged pixel only once per wave); with 64 couts the second accumulator pair costs a
    // resident workgroup (2 instead of 3 per CU) and it is a wash on big grids -- but on grids of 1024-8192 workgroups
    // 2 x 256 slots divide the grid evenly where 3 x 256 leave a ragged last round: +2-7 % there (tools/ab_lib.py,
    // MRDIS_DEBUG_BM: 0 never, 1 BN=32 layers, 2 all, 3 = default policy).  Only where the hoisted staging still fits
    // its register arrays.
    {
        int want = 3;
        if (mrdis_opt(MRDIS_OPT_BM) >= 0) want = (int)mrdis_opt(MRDIS_OPT_BM);
        const bool cand = p.prefetch == 1 && KC == 16 && ((nblk >= 4096 && (want == 1 || want == 3) && BN == 32) || (want == 2 && nblk >= 1024) || (want == 3 && BN == 64 && nblk >= 1024 && nblk <= 8192));
        if (cand) {
            TapConvParams q = p;
            const TileChoice t2 = choose_tile(q.N, q.A, q.B, 256);
            q.NB = t2.NB; q.TH = t2.TH; q.TW = t2.TW;
            q.TinH = (q.TH - 1) * q.is + (dh_max - q.dh_min) + 1;
            q.TinW = (q.TW - 1) * q.is + (dw_max - q.dw_min) + 1;
            q.tilesA = mrdis_cdiv(q.A, q.TH); q.tilesB = mrdis_cdiv(q.B, q.TW); q.tilesN = mrdis_cdiv(q.N, q.NB);
            const long long npix2 = (long long)q.NB * q.TinH * q.TinW;
            const long long nblk2 = (long long)q.tilesA * q.tilesB * q.tilesN * q.coTiles;
            const size_t lds2 = tapconv_lds(q, KC, BN, 256);
            if (q.NB * q.TH * q.TW == 256 && npix2 * (KC / 4) <= 6 * 256 && lds2 <= LDS_MAX) { p = q; nblk = nblk2; lds = lds2; BM = 256; }
        }
    }
    if (defer) { defer->p = p; defer->KC = KC; defer->BN = BN; defer->BM = BM; defer->nblk = (int)nblk; defer->lds = lds; defer->set = true; return MRDIS_OK; }
    // half-filled small-map tiles: two waves per 32 x 32 block (tapconv_body, SK = 2); the partials need 8 KB of the filter slab
    const bool split_k = small_map && BN == 32 && KC >= 16 && BM == 128 && p.prefetch == 1 && p.NB * p.TH * p.TW <= 64 &&
                         (size_t)p.ntaps * KC * BN >= 2048 && !mrdis_opt(MRDIS_OPT_NOW16);
#define TC_CASE(kc, bn) if (KC == kc && BN == bn) return launch_tapconv_t<kc, bn>(p, lds, (int)nblk, BM, s, split_k)
    TC_CASE(4, 32); TC_CASE(4, 64);
    TC_CASE(8, 32); TC_CASE(8, 64);
    TC_CASE(16, 32); TC_CASE(16, 64);
    TC_CASE(32, 32);
#undef TC_CASE
    return MRDIS_EUNSUPPORTED;
}

static int check_conv_geom(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad,
                           int* Ho, int* Wo) {
    if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return MRDIS_EINVAL;
    if (kh < 1 || kw < 1 || kh * kw > MRDIS_MAX_TAPS) return MRDIS_EUNSUPPORTED;
    if (stride != 1 && stride != 2) return MRDIS_EUNSUPPORTED;
    if (pad < 0 || pad >= kh || pad >= kw) return MRDIS_EUNSUPPORTED;
    *Ho = (H + 2 * pad - kh) / stride + 1;
    *Wo = (W + 2 * pad - kw) / stride + 1;
    if (*Ho <= 0 || *Wo <= 0) return MRDIS_EINVAL;
    return MRDIS_OK;
}


// =========================================================================== Cin = 4 direct conv (north star)
// 3x3 / stride 1 / pad 1 convolution of a 4-channel map (the `si_layers` of every SPADE block,
// reference model.py:2433; SURVEY.md 8d: x (32,4,240,240) -> y (32,32,240,240)).  K = 9 taps x 4
// channels = 36 = 18 MFMA k-steps exactly, arithmetic intensity 16 FLOP/B: HBM and the fp32 MFMA
// pipe are both near their floors, so nothing may be wasted:
//   * no LDS, no barriers: a wave owns a strip of 32 output positions; lane (m, half) fetches its
//     own 3x3 window as nine 8-byte loads (channels 2*half, 2*half+1) -- the two half-waves
//     together read each 16-byte pixel exactly once per tap, re-reads of the halo are L1 hits;
//   * the whole filter lives in registers (18 * NS VGPRs) for the lifetime of the wave;
//   * persistent waves walk the strips with a grid stride, prefetching TWO strips ahead: tap t of
//     strip i+2 is loaded into the register pair the MFMAs of tap t of strip i have just read;
//   * all global traffic goes through per-image buffer descriptors with 32-bit offsets: a tap outside
//     the image (or a position of a ragged strip) gets an offset beyond `num_records`, so the hardware
//     range check returns zeros / drops the store -- zero padding and ragged edges cost no branch,
//     no select at use and no exec masking;
//   * the strip loop is STRAIGHT-LINE code on purpose: every conditional block inside it is a
//     control-flow join at which hipcc's wait-count pass forgets which loads are pending and falls back
//     to `s_waitcnt vmcnt(0)` -- which also waits for the stores of the previous strip (vmcnt retires
//     in order), serialising the write stream against the read stream;
//   * accumulators start at the bias (C operand of the first MFMA, kept in its own accumulator set);
//   * D[position][cout]: lane (m, half) owns cout m of 16 positions, so every store instruction writes
//     two complete 128-byte lines (32 couts x 4 B for two positions).  The swapped layout (a lane owns 16
//     couts of one position, four 16-byte stores per strip) issues a quarter of the instructions but
//     reaches L2 as 32-byte partial-line requests -- 7.4 M requests per launch instead of 1.8 M, and the L2
//     request rate, not HBM, then bounds the kernel (measured: 58.6 -> 52.2 us in-cache, 106 -> 89 us
//     beyond the 256 MB Infinity Cache).
// cache policy of the output stores: bit 1 = NT (non-temporal).  The 236-268 MB output stream is not re-read by this
// kernel and is as large as the Infinity Cache; marking it streaming keeps the halo rows of x resident: 53.5 -> 49.3 us at
// 240x240, 68.0 -> 58.3 us at 256x256 (tools/ab_lib.py, 8 rounds).  SC0 / SC1 on top made no further difference.
#define C4_STORE_NT 2
struct C4Params {
    const float* x; const float* w; const float* bias; float* y;
    int N, H, W, ldx, Co, ldy;
    int TW, TH;                  // strip shape: TW x TH = 32 positions (32x1, 16x2 or 8x4)
    int tilesW, tilesH;          // per image
    long long ntiles;
    int lrelu;
    int flip;                    // taps in reverse order: the stride-1 data gradient of a C -> 4 layer is this same convolution
    int wrows;                   // channel rows per tap of the filter in memory: 4, or 16 for the zero-padded [9][16][Co] layout the mixing
                                 // launch writes under bf16 storage (rows 4..15 are zero and never read)
};
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define C4_OOB 0x40000000u       // any offset >= this is outside every descriptor this kernel builds

// OBF16: the output is a bf16 view (MRDIS_DT_XF32_YBF16: the si_layers under `compute_dtype: bf16` read the fp32 anatomy map and open a
// bf16 stretch).  A lane owns ONE cout, so a bf16 store of its own would be 2 bytes; adjacent lanes exchange one value per register pair
// (DPP quad_perm 1,0,3,2) and each stores a packed cout PAIR: even lanes (m, m+1) of position r, odd lanes (m-1, m) of position r + 1 --
// half the store instructions of the fp32 form, every one of them 4 bytes per lane.
// SPLIT6 (fp32 output, option split6): the bf16 matrix pipe at fp32 accuracy -- BOTH operands as three bf16 terms (v = hi + mid + lo, 3 x 8 mantissa bits) and
// the six products of order <= 2, x_h w_h + x_m w_h + x_h w_m + x_l w_h + x_h w_l + x_m w_m (what is dropped is below 2^-23 of a product): 54 groups of four
// channels = 14 k-steps of 32 cycles against 18 fp32 MFMAs of 64.  Same ownership of whole taps by the half-waves as the two-term form below; per tap the
// x side of its three k-steps is (h m) (h l) (h m) against the filter's (wh wh) (wm wh) (wl wm); tap 8: (h m) (h l) against (wh wh) (wm wh) in the lower and
// (wl wm) (0 0) in the upper half-wave.
template <int NS, bool LRELU, bool FULL, bool OBF16, bool SPLIT6 = false>   // FULL: the strips tile the image exactly (no per-store position checks)
__device__ __forceinline__ void c4conv_body(const C4Params& p) {
    static_assert(!(OBF16 && SPLIT6), "SPLIT6 is the fp32-output form");
    constexpr bool BFP = OBF16 || SPLIT6;              // the product runs on the bf16 matrix pipe
    const int lane = threadIdx.x & 63, half = lane >> 5, m = lane & 31;
    const unsigned ty = m / p.TW, tx = m - ty * p.TW;
    // wave-uniform strip bookkeeping lives in SGPRs (readfirstlane makes the uniformity provable)
    const int wave_in_blk = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)gridDim.x * 4;
    const int ntiles = (int)p.ntiles;
    const int cot = blockIdx.y;                         // 32*NS couts per y-slice of the grid
    // filter -> registers: b[t][j][ns] = w[t][2*half + j][co]
    // OBF16 (the layer opens / closes a bf16 stretch): the product runs on the bf16 matrix pipe, v_mfma_f32_32x32x16_bf16 -- SEVEN MFMAs of 32 cycles per
    // strip instead of eighteen fp32 ones of 64 (the fp32 form is bound by them: 0.65 of the fp32 MFMA peak at the headline shape).  Both fp32 operands
    // are carried to 16 mantissa bits as two bf16 terms, v = hi + mid (hi = bf16(v), mid = bf16(v - hi)), and the three products that matter are
    // summed: x_hi w_hi + x_mid w_hi + x_hi w_mid (the fourth is below 2^-18 of the first) -- 2^-16 relative against the fp32 product, far under the
    // bf16 rounding of the output.  k runs over (tap, product, channel): 27 groups of four channels + 1 of zeros = 7 k-steps of 32.  A half-wave owns
    // whole taps: the lower one taps 0-3 and the (hi, hi) (mid, hi) products of tap 8, the upper one taps 4-7 and tap 8's (hi, mid) -- so a lane
    // loads FIVE taps of 16 bytes (its four + tap 8) instead of nine of 8, and the x side of the seven k-steps is the same expression of those five
    // loads in both half-waves: (L0h L0m) (L0h L1h) (L1m L1h) (L2h L2m) (L2h L3h) (L3m L3h) (L4h L4m); the filter side, set up once, is
    // (wh wh) (wm wh) (wh wm) ... and for tap 8 (wh wh) in the lower, (wm 0) in the upper half-wave.
    typedef __bf16 c4_bf16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 c4_bf16x4 __attribute__((ext_vector_type(4)));
    constexpr int NTAPL = BFP ? 5 : 9;                 // tap loads per strip
    constexpr int NKS = SPLIT6 ? 14 : (OBF16 ? 7 : 9); // pipeline slots (k-steps | taps) per strip
    float b[BFP ? 1 : 9][2][NS];
    c4_bf16x8 bw[BFP ? NKS : 1][NS];
    if (SPLIT6) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
            const int co = (cot * NS + ns) * 32 + m;
            const int coc = co < p.Co ? co : p.Co - 1;
            c4_bf16x4 wh[5], wm[5], wl[5];             // local tap l: tap 4 half + l (l < 4), tap 8 (l = 4)
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                const int tap = l < 4 ? 4 * half + l : 8;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = co < p.Co ? p.w[((p.flip ? 8 - tap : tap) * p.wrows + c) * p.Co + coc] : 0.f;
                    const __bf16 h = (__bf16)v; const float r1 = v - (float)h; const __bf16 mi = (__bf16)r1;
                    wh[l][c] = h; wm[l][c] = mi; wl[l][c] = (__bf16)(r1 - (float)mi);
                }
            }
            const c4_bf16x4 z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            c4_bf16x4 seq[28];
#pragma unroll
            for (int l = 0; l < 4; ++l) { seq[6 * l] = wh[l]; seq[6 * l + 1] = wh[l]; seq[6 * l + 2] = wm[l]; seq[6 * l + 3] = wh[l]; seq[6 * l + 4] = wl[l]; seq[6 * l + 5] = wm[l]; }
            seq[24] = half ? wl[4] : wh[4]; seq[25] = half ? wm[4] : wh[4]; seq[26] = half ? z4 : wm[4]; seq[27] = half ? z4 : wh[4];
#pragma unroll
            for (int s14 = 0; s14 < 14; ++s14)
#pragma unroll
                for (int c = 0; c < 4; ++c) { bw[s14][ns][c] = seq[2 * s14][c]; bw[s14][ns][4 + c] = seq[2 * s14 + 1][c]; }
        }
    } else
    if (OBF16) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
            const int co = (cot * NS + ns) * 32 + m;
            const int coc = co < p.Co ? co : p.Co - 1;
            c4_bf16x4 wh[5], wm[5];                    // local tap l: tap 4 half + l (l < 4), tap 8 (l = 4)
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                const int tap = l < 4 ? 4 * half + l : 8;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = co < p.Co ? p.w[((p.flip ? 8 - tap : tap) * p.wrows + c) * p.Co + coc] : 0.f;
                    const __bf16 h = (__bf16)v;
                    wh[l][c] = h; wm[l][c] = (__bf16)(v - (float)h);
                }
            }
            const c4_bf16x4 z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            // group sequence of the filter side: per local tap l < 4: wh wh wm (against x: h m h); tap 8: lower half wh wh, upper half wm 0
            c4_bf16x4 seq[14];
#pragma unroll
            for (int l = 0; l < 4; ++l) { seq[3 * l] = wh[l]; seq[3 * l + 1] = wh[l]; seq[3 * l + 2] = wm[l]; }
            seq[12] = half ? wm[4] : wh[4]; seq[13] = half ? z4 : wh[4];
#pragma unroll
            for (int s7 = 0; s7 < 7; ++s7)
#pragma unroll
                for (int c = 0; c < 4; ++c) { bw[s7][ns][c] = seq[2 * s7][c]; bw[s7][ns][4 + c] = seq[2 * s7 + 1][c]; }
        }
    } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                const int co = (cot * NS + ns) * 32 + m;
                const int coc = co < p.Co ? co : p.Co - 1;
                const float v = p.w[((p.flip ? 8 - t : t) * p.wrows + 2 * half + j) * p.Co + coc];
                b[t][j][ns] = co < p.Co ? v : 0.f;
            }
    }
    // bias in the accumulator layout (every register of a lane belongs to cout m).  Passing it once through
    // the matrix pipe (0*0 + bias) gives the compiler an accumulator-class value it can keep resident.
    // (OBF16: the bias is ONE value per lane, added where the accumulator is rounded to bf16 -- a resident accumulator set per 32 couts is 16 VGPRs the
    //  seven-MFMA form cannot spare; its first MFMA takes the inline constant 0 as C)
    f32x16 bv[NS];
    float bias_m[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int co = (cot * NS + ns) * 32 + m;
        const float v = (p.bias != nullptr && co < p.Co) ? p.bias[co < p.Co ? co : 0] : 0.f;
        bias_m[ns] = v;
        f32x16 t16;
#pragma unroll
        for (int r = 0; r < 16; ++r) t16[r] = BFP ? 0.f : v;
        if (BFP) bv[ns] = t16; else bv[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.f, 0.f, t16, 0, 0, 0);
    }
    const int tpi = p.tilesW * p.tilesH;
    // strip index -> (n, th, tw), advanced incrementally by the grid stride (no divisions in the loop)
    const int d_n = nwaves / tpi, d_rem = nwaves - d_n * tpi, d_th = d_rem / p.tilesW, d_tw = d_rem - d_th * p.tilesW;
    // XCD-aware: workgroups b and b + 8 share an XCD (and its L2); give each XCD a contiguous run of strips per grid-stride round so
    // that the strips above / below a strip (whose halo rows it re-reads) are fetched into the same L2
    int tile = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x) * 4 + wave_in_blk;
    if (tile >= ntiles) return;
    auto advance = [&](int& n, int& th, int& tw) {
        tw += d_tw; if (tw >= p.tilesW) { tw -= p.tilesW; th += 1; }
        th += d_th; if (th >= p.tilesH) { th -= p.tilesH; n += 1; }
        n += d_n;
    };
    // geometry in bytes (host guarantees: image bytes < C4_OOB, every 24-bit multiply operand < 2^24)
    const unsigned pix = 4u * p.ldx, rowbytes = pix * (unsigned)p.W, imgbytes = rowbytes * (unsigned)p.H;
    const unsigned opix = (OBF16 ? 2u : 4u) * p.ldy, oimgbytes = opix * (unsigned)p.W * (unsigned)p.H;
    const unsigned lane_off = BFP ? 0u : 8u * half;    // (bf16 matrix pipe: a lane loads all four channels of its tap)
    // OBF16: the cout pair this lane stores; odd lanes store the pair of the NEXT position of the strip row (P(r) + 1 for even r never leaves
    // the row), which is one output pixel further -- a lane constant, so the per-register part stays a wave-uniform scalar operand as in fp32
    const unsigned st_lane = OBF16 ? 2u * (m & ~1u) + 64u * NS * cot + ((m & 1) ? opix : 0u)
                                   : 4u * m + 128u * NS * cot;             // cout m of this y-slice
    const unsigned uH = p.H, uW = p.W;
    // timing-only builds (-DC4_ABL_NOLOAD / _NOSTORE / _NOMFMA): a zero-record descriptor drops the traffic
    // while the instruction stream and the waits stay
#ifdef C4_ABL_NOLOAD
    const unsigned xrec = 0;
#else
    const unsigned xrec = imgbytes;
#endif
#ifdef C4_ABL_NOSTORE
    const unsigned yrec = 0;
#else
    const unsigned yrec = oimgbytes;
#endif
    // (readfirstlane: the strip bookkeeping is wave-uniform, but in the bf16-output instantiation the allocator moved the image index into a VGPR
    //  and every buffer instruction became a waterfall loop -- one readfirstlane / compare / branch round per load and store: 88 vs 57 us)
    auto uni64 = [](unsigned long long a) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    };
    auto x_desc = [&](int n) {
        const unsigned long long a = uni64((unsigned long long)(uintptr_t)(p.x + (size_t)n * (imgbytes / 4)));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)a, 0, __builtin_amdgcn_readfirstlane((int)xrec), 0x00020000);
    };
    auto y_desc = [&](int n) {
        const unsigned long long a = uni64((unsigned long long)(uintptr_t)((char*)p.y + (size_t)n * oimgbytes));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)a, 0, __builtin_amdgcn_readfirstlane((int)yrec), 0x00020000);
    };
    // 9 tap byte offsets of a strip: 3 row bases x 3 column offsets, each either valid or C4_OOB.
    auto strip_offsets = [&](int th, int tw, unsigned (&voff)[NTAPL]) {
        const unsigned h = (unsigned)(th * p.TH) + ty, w_ = (unsigned)(tw * p.TW) + tx;
        const unsigned r1 = __umul24(h, rowbytes) + lane_off, c1 = __umul24(w_, pix);
        unsigned rb[3], cb[3];
        rb[0] = (h - 1u < uH) ? r1 - rowbytes : C4_OOB;
        rb[1] = (h < uH) ? r1 : C4_OOB;
        rb[2] = (h + 1u < uH) ? r1 + rowbytes : C4_OOB;
        cb[0] = (w_ - 1u < uW) ? c1 - pix : C4_OOB;
        cb[1] = (w_ < uW) ? c1 : C4_OOB;
        cb[2] = (w_ + 1u < uW) ? c1 + pix : C4_OOB;
        if (BFP) {                                     // load l of a lane: tap 4 half + l (l < 4), tap 8 (l = 4)
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const int t0 = l, t1 = 4 + l;
                voff[l] = half ? rb[t1 / 3] + cb[t1 % 3] : rb[t0 / 3] + cb[t0 % 3];
            }
            voff[4] = rb[2] + cb[2];
        } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) voff[(3 * r + c) % NTAPL] = rb[r] + cb[c];
        }
    };
    // accumulator register r of a lane is strip position P(r) + 4*half, P(r) = (r&3) + 8*(r>>2): row P/TW,
    // column P%TW (+4*half never crosses a strip row: TW >= 8).  The per-register part of the store offset is
    // wave-uniform and goes into the scalar offset operand; the lane part is the strip origin + half + cout.
    auto store_offset = [&](int th, int tw) -> unsigned {             // strip origin + lane part, or OOB
        const unsigned o = __umul24(__umul24((unsigned)(th * p.TH), uW) + (unsigned)(tw * p.TW) + 4u * half, opix) + st_lane;
        return o;
    };
    auto load_tap = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, auto& dst) {
        if constexpr (BFP) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
            dst = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else {
            const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, 0, 0);
            dst = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
        }
    };
    // a loaded tap as two bf16 terms: hi = bf16(x) (round to nearest even), mid = bf16(x - hi)
    auto split_tap = [&](const float4& x, c4_bf16x4& hi, c4_bf16x4& mid) {
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const __bf16 h = (__bf16)xv[c]; hi[c] = h; mid[c] = (__bf16)(xv[c] - (float)h); }
    };
    auto split_tap3 = [&](const float4& x, c4_bf16x4& hi, c4_bf16x4& mid, c4_bf16x4& lo) {
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const __bf16 h = (__bf16)xv[c]; const float r1 = xv[c] - (float)h; const __bf16 mi = (__bf16)r1;
            hi[c] = h; mid[c] = mi; lo[c] = (__bf16)(r1 - (float)mi);
        }
    };
    auto cat8 = [](const c4_bf16x4& lo, const c4_bf16x4& hi) -> c4_bf16x8 { return c4_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; };
    unsigned lane_ok_off[NS];                          // C4_OOB for lanes whose cout does not exist
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) lane_ok_off[ns] = ((cot * NS + ns) * 32 + m < p.Co) ? 0u : C4_OOB;
    auto store_group = [&](const f32x16 (&acc)[NS], __amdgpu_buffer_rsrc_t rs, unsigned so, int th, int tw, int g) {
        if (OBF16) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const bool odd = (m & 1) != 0;
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                const int r0 = 4 * g + k, r1 = r0 + 1;                                  // P(r1) = P(r0) + 1: the next position of the strip row
                const int P0 = (r0 & 3) + 8 * (r0 >> 2);
                const int pty0 = P0 / p.TW, ptx0 = P0 - pty0 * p.TW;                     // wave-uniform
                const unsigned soff = ((unsigned)pty0 * uW + (unsigned)ptx0) * opix;      // scalar operand (the odd lanes' + 1 pixel is in st_lane)
                unsigned vo = so;
                if (!FULL) {
                    const unsigned h = (unsigned)(th * p.TH + pty0), w_ = (unsigned)(tw * p.TW + ptx0) + (odd ? 1u : 0u) + 4u * half;
                    vo = ((int)(h < uH) & (int)(w_ < uW)) ? so : C4_OOB;
                }
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    float a0 = acc[ns][r0] + bias_m[ns], a1 = acc[ns][r1] + bias_m[ns];
                    if (LRELU) { a0 = fmaxf(a0, 0.2f * a0); a1 = fmaxf(a1, 0.2f * a1); }
                    const float send = odd ? a0 : a1;                                  // what the neighbour's store needs from this lane
                    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));
                    bf16x2_t pk;
                    pk[0] = (__bf16)(odd ? recv : a0);                                 // the lower cout of the pair
                    pk[1] = (__bf16)(odd ? a1 : recv);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk), rs, (int)((vo | lane_ok_off[ns]) + 64u * ns), (int)soff, C4_STORE_NT);
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 4 * g + k;
            const int P = (r & 3) + 8 * (r >> 2);
            const int pty = P / p.TW, ptx = P - pty * p.TW;                       // wave-uniform
            const unsigned soff = ((unsigned)pty * uW + (unsigned)ptx) * opix;       // scalar operand
            unsigned vo = so;
            if (!FULL) {
                const unsigned h = (unsigned)(th * p.TH + pty), w_ = (unsigned)(tw * p.TW + ptx) + 4u * half;
                vo = ((int)(h < uH) & (int)(w_ < uW)) ? so : C4_OOB;
            }
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                float v = SPLIT6 ? acc[ns][r] + bias_m[ns] : acc[ns][r];
                if (LRELU) v = fmaxf(v, 0.2f * v);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)((vo | lane_ok_off[ns]) + 128u * ns), (int)soff, C4_STORE_NT);
            }
        }
    };

    // Register pipeline: two strip buffers alternate (X0: even strips of this wave, X1: odd); iteration i
    // runs the MFMAs of tap t from its buffer and immediately refills that pair with tap t of strip i+2.
    // Two accumulator sets alternate as well (the set of strip i-1 is stored while strip i computes).
    // Loads are issued BEFORE the stores of an iteration slot, so waiting for a load never waits for a
    // younger store (vmcnt retires in order).
    typedef typename std::conditional<BFP, float4, float2>::type xtap_t;
    xtap_t X0[NTAPL], X1[NTAPL];
    int cn = tile / tpi, c_rem = tile - cn * tpi, cth = c_rem / p.tilesW, ctw = c_rem - cth * p.tilesW;
    int ln = cn, lth = cth, ltw = ctw;                  // strip the next loads belong to
    int ltile = tile;
    f32x16 accA[NS], accB[NS];
    unsigned pso = C4_OOB; int pn = 0, pth = 0, ptw = 0; // store offset / image / strip of the previous strip (none yet:
                                                        // the first stores are dropped by the range check)
    // One pipeline step.  PRO (prologue) steps issue exactly the memory instructions of a real step -- nine
    // loads and four (dropped) stores -- and no MFMAs, so the wait-count pass sees the same number of
    // outstanding operations on the loop-entry path as on the back edge and keeps the two-strip prefetch
    // distance instead of clamping it to the shorter path.
    auto step = [&](auto pro, f32x16 (&acc)[NS], const f32x16 (&prev)[NS], xtap_t (&X)[NTAPL]) {
        constexpr bool PRO = decltype(pro)::value;
        const bool have_l = ltile < ntiles;                               // prefetch past the end -> strip 0
        unsigned vo[NTAPL];
        strip_offsets(have_l ? lth : 0, have_l ? ltw : 0, vo);
        const __amdgpu_buffer_rsrc_t rsx = x_desc(have_l ? ln : 0);
        const __amdgpu_buffer_rsrc_t rsy = y_desc(pn);
        const unsigned cso = PRO ? C4_OOB : store_offset(cth, ctw);
        if constexpr (SPLIT6) {
            // a tap is split when its first k-step comes up and its raw registers are refilled (strip i + 2) behind its last one: one tap's three terms live at a
            // time (all five up front: 192 VGPRs, two waves per SIMD)
            c4_bf16x4 xh, xm, xl;
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 14; ++t) {
                const int l = t / 3;                   // k-steps 3 l .. 3 l + 2 belong to local tap l (12, 13: tap 8)
                if (!PRO) {
                    if (t % 3 == 0) split_tap3(X[l], xh, xm, xl);
                    const c4_bf16x8 a8 = (t % 3 == 1) ? cat8(xh, xl) : cat8(xh, xm);
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bw[t][ns], t == 0 ? zero16 : acc[ns], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t == 2 || t == 5 || t == 8 || t == 11 || t == 13) load_tap(rsx, vo[l], X[l]);
                if (t == 6 || t == 7 || t == 9 || t == 10) store_group(prev, rsy, pso, pth, ptw, t < 8 ? t - 6 : t - 7);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        if constexpr (OBF16) {
            c4_bf16x4 xh[5], xm[5];
            if (!PRO) {
#pragma unroll
                for (int l = 0; l < 5; ++l) split_tap(X[l], xh[l], xm[l]);
            }
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                if (!PRO) {
                    // x side of k-step t (see the top of the kernel)
                    const c4_bf16x8 a8 = t == 0 ? cat8(xh[0], xm[0]) : t == 1 ? cat8(xh[0], xh[1]) : t == 2 ? cat8(xm[1], xh[1]) : t == 3 ? cat8(xh[2], xm[2])
                                       : t == 4 ? cat8(xh[2], xh[3]) : t == 5 ? cat8(xm[3], xh[3]) : cat8(xh[4], xm[4]);
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, bw[t][ns], t == 0 ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : acc[ns], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t >= 2) load_tap(rsx, vo[t - 2], X[t - 2]);           // (the split terms above are what the MFMAs read: the raw taps are free)
                if (t >= 3) store_group(prev, rsy, pso, pth, ptw, t - 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!PRO) {
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
#ifdef C4_ABL_NOMFMA
                    if (t == 0) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[t][0][ns], X[t].x, bv[ns], 0, 0, 0);
                    else asm volatile("" :: "v"(X[t].x), "v"(X[t].y));
#else
                    acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[t].x, b[t][0][ns], t == 0 ? bv[ns] : acc[ns], 0, 0, 0);
                    acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[t].y, b[t][1][ns], acc[ns], 0, 0, 0);
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_tap(rsx, vo[t], X[t]);
            if (t >= 5) store_group(prev, rsy, pso, pth, ptw, t - 5);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (!PRO) {
            pso = cso; pn = cn; pth = cth; ptw = ctw;
            advance(cn, cth, ctw);
        }
        advance(ln, lth, ltw);
        ltile += nwaves;
    };
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) { accA[ns] = bv[ns]; accB[ns] = bv[ns]; }
    step(std::true_type{}, accA, accB, X0);             // strip 0 -> X0
    step(std::true_type{}, accB, accA, X1);             // strip 1 -> X1
    bool last_is_A = true;
    while (true) {
        step(std::false_type{}, accA, accB, X0);
        tile += nwaves; last_is_A = true;
        if (tile >= ntiles) break;
        step(std::false_type{}, accB, accA, X1);
        tile += nwaves; last_is_A = false;
        if (tile >= ntiles) break;
    }
    const __amdgpu_buffer_rsrc_t rsy = y_desc(pn);
    if (last_is_A) {
#pragma unroll
        for (int g = 0; g < 4; ++g) store_group(accA, rsy, pso, pth, ptw, g);
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) store_group(accB, rsy, pso, pth, ptw, g);
    }
}
template <int NS, bool LRELU, bool FULL, bool OBF16 = false>
__global__ __launch_bounds__(256) void c4conv_kernel(const C4Params p) { c4conv_body<NS, LRELU, FULL, OBF16>(p); }
template <bool LRELU, bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void c4conv_split6_kernel(const C4Params p) { c4conv_body<1, LRELU, FULL, false, true>(p); }
// the 32-cout bf16-output form on exact strips (the si_layers' forward at 256x256): left alone the allocator takes 95 + 48 registers,
// one wave per SIMD fewer than the fp32 form's 88 + 32 -- and the kernel lives on waves in flight (72 vs 51 us).  Pinned to three waves per SIMD (round 5: the
// two-term form needs 149 registers); the persistent grid stays at four workgroups per CU (run_c4conv: measured).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void c4conv_obf16_kernel(const C4Params p) { c4conv_body<1, false, true, true>(p); }

static bool c4_eligible(const float* x, int ldx, int ldy, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int obytes = 4) {
    if (!(Ci == 4 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && (ldx % 2 == 0) && (((uintptr_t)x & 7) == 0) && Co >= 16)) return false;
    // 32-bit offsets against per-image buffer descriptors, 24-bit multiplies
    const long long xin = 4LL * ldx * W * H, yout = (long long)obytes * ldy * W * H;
    return xin < 0x40000000LL && yout < 0x40000000LL && 4LL * ldx * W < (1 << 24) && (long long)H * W < (1 << 24) && (long long)obytes * ldy < (1 << 24);
}

static int run_c4conv(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                      int N, int H, int W, int Co, int epilogue, hipStream_t s, int flip = 0, int wrows = 4, bool obf16 = false) {
    C4Params p{};
    p.flip = flip; p.wrows = wrows;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.N = N; p.H = H; p.W = W; p.ldx = ldx; p.Co = Co; p.ldy = ldy;
    p.lrelu = (epilogue & MRDIS_EPI_LRELU) ? 1 : 0;
    // strip shape: 32x1 unless a 16x2 strip wastes fewer positions
    const double u32 = (double)W / (mrdis_cdiv(W, 32) * 32.0);
    const double u16 = ((double)W / (mrdis_cdiv(W, 16) * 16.0)) * ((double)H / (mrdis_cdiv(H, 2) * 2.0));
    double best = u32; p.TW = 32; p.TH = 1;
    if (u16 > best + 1e-9) { best = u16; p.TW = 16; p.TH = 2; }
    if (W <= 8) { p.TW = 8; p.TH = 4; }
    if (mrdis_opt(MRDIS_OPT_C4_TW) > 0) { p.TW = (int)mrdis_opt(MRDIS_OPT_C4_TW); p.TH = 32 / p.TW; }
    p.tilesW = mrdis_cdiv(W, p.TW); p.tilesH = mrdis_cdiv(H, p.TH);
    p.ntiles = (long long)N * p.tilesW * p.tilesH;
    const int co32 = mrdis_cdiv(Co, 32);
    // option split6: 1 (default): here for <= 32 couts (wider layers would run as two 32-cout slices: 107 -> 123 us on 4 <- 64 at 256x256), and the C -> 4 kernel |
    // 2: here only, every width | 3: the C -> 4 kernel only | 4: both, every width | 5 / 6: only the 32 -> 16 forward (mrdis_c16.hip) / its weight gradient (mrdis_wgrad16.hip), both also under 1 | 7: only its data gradient (16 -> 32, mrdis_c16.hip; not under 1: no gain in the step) | 0: fp32 MFMA everywhere
    const long long s6 = mrdis_opt(MRDIS_OPT_SPLIT6);
    const bool split6 = !obf16 && ((s6 == 1 && co32 == 1) || s6 == 2 || s6 == 4);        // six bf16 products per fp32 product; 32 couts per workgroup (14 x 4 filter registers per 32)
    const int NS = (co32 % 2 == 0 && co32 >= 2 && !split6) ? 2 : 1;
    const int ny = co32 / NS;
    long long blocks = (p.ntiles + 3) / 4;
    // persistent grid: exactly the number of workgroups the chip holds at once (one wave of workgroups,
    // no tail round); residency is queried once per instantiation.
    // Grid policy, measured (tools/c4_grid_probe.py, round 6): FOUR workgroups per CU for every form.  The six-product and bf16-output forms are pinned to
    // three waves per SIMD, so only three of the four are resident at a time -- sizing the grid by their own occupancy (768 workgroups) was slower: 59.6 vs
    // 53.6 us at 256x256, 52.7 vs 51.8 us at 240x240 (the fourth workgroup of a CU starts as the first drains and evens out the tail).  Option c4_grid = k
    // overrides the number per CU.
    static int occ[3] = {0, 0, 0}, ncu = 0;
    const int oi = NS;
    if (!occ[oi]) {
        int o = 0;
        if (NS == 2) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, c4conv_kernel<2, false, true>, 256, 0);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, c4conv_kernel<1, false, true>, 256, 0);
        occ[oi] = o > 0 ? o : 2;
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const long long per_cu = mrdis_opt(MRDIS_OPT_C4_GRID) > 0 ? mrdis_opt(MRDIS_OPT_C4_GRID) : occ[oi];
    long long cap = (long long)ncu * per_cu / ny; if (cap < ncu) cap = ncu;
    if (blocks > cap) blocks = cap;
    mrdis_opt_note(MRDIS_OPT_C4_BLOCKS, blocks);
    const bool fast = (W % p.TW == 0) && (H % p.TH == 0);      // FULL: strips tile the image exactly
    const dim3 grid((int)blocks, ny);
#define C4_LAUNCH(ns, lr, fa) MRDIS_LAUNCH((c4conv_kernel<ns, lr, fa>), grid, dim3(256), 0, s, p)
    if (split6) {
        mrdis_count(MRDIS_CNT_SPLIT6_C4);
        if (p.lrelu) { if (fast) MRDIS_LAUNCH((c4conv_split6_kernel<true, true>), grid, dim3(256), 0, s, p); else MRDIS_LAUNCH((c4conv_split6_kernel<true, false>), grid, dim3(256), 0, s, p); }
        else { if (fast) MRDIS_LAUNCH((c4conv_split6_kernel<false, true>), grid, dim3(256), 0, s, p); else MRDIS_LAUNCH((c4conv_split6_kernel<false, false>), grid, dim3(256), 0, s, p); }
    } else
    if (obf16) {       // the si_layers' forward (flip = 0) and the C <- 4 data gradient (flip = 1: run time); no LeakyReLU follows either
        if (NS == 2) { if (fast) MRDIS_LAUNCH((c4conv_kernel<2, false, true, true>), grid, dim3(256), 0, s, p); else MRDIS_LAUNCH((c4conv_kernel<2, false, false, true>), grid, dim3(256), 0, s, p); }
        else { if (fast) MRDIS_LAUNCH(c4conv_obf16_kernel, grid, dim3(256), 0, s, p); else MRDIS_LAUNCH((c4conv_kernel<1, false, false, true>), grid, dim3(256), 0, s, p); }
    } else
    if (NS == 2) {
        if (p.lrelu) { if (fast) C4_LAUNCH(2, true, true); else C4_LAUNCH(2, true, false); }
        else { if (fast) C4_LAUNCH(2, false, true); else C4_LAUNCH(2, false, false); }
    } else {
        if (p.lrelu) { if (fast) C4_LAUNCH(1, true, true); else C4_LAUNCH(1, true, false); }
        else { if (fast) C4_LAUNCH(1, false, true); else C4_LAUNCH(1, false, false); }
    }
#undef C4_LAUNCH
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// fused Winograd F(2x2, 3x3) kernel (mrdis_wino.hip)
int mrdis_run_wino(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                   int N, int H, int W, int Ci, int Co, int flip, int lrelu, hipStream_t s);
// MRDIS_WINO: 0 never | 1 (default) measured policy | 2 wherever the kernel applies.  Policy (tools/layer_bench.py, B = 32
// layer zoo): Winograd wins for Cout >= 32 and Cin >= 16 once the grid fills the chip (>= 256 workgroups): 1.13x on
// 32 -> 32, 1.3-1.5x on the 64..512-channel layers; a 16-cout layer wastes half of its 32-wide cout tile (slower).
// software-pipelined variant for Cout > 32 (mrdis_wino2.hip); option wino_pipe = 0 keeps the phase-by-phase kernel everywhere
int mrdis_run_wino2(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int N, int H, int W, int Ci, int Co, int flip, int lrelu, hipStream_t s, const float* u_img);
// F(4x4, 3x3) (mrdis_wino4.hip): takes the layers whose filter image is of format 4 (mrdis_wino_u_format; the image carries the flip, so
// forward and data gradient are the same call); MRDIS_EUNSUPPORTED where its shape limits / grid test decline
int mrdis_run_wino4(const float* x, int ldx, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu,
                    hipStream_t s, const float* u_img);
#include "mrdis_wino4.h"
int mrdis_run_wino4n(const float* x, int ldx, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu,
                     hipStream_t s, const float* u_img);
int mrdis_run_wino4r(const float* x, int ldx, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu,
                     hipStream_t s, const float* u_img);     // the register-fed form of the same layers (mrdis_wino4r.hip)
int mrdis_run_wino4_spade(const float* x, int ldx, const float* bias, const float* z, int ldz, const float* mean, const float* rstd,
                          float* mix, int ldmix, float* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s, const float* u_img);
// u_fmt: the format the image u_img was BUILT in (2 | 4 | 5: mrdis_wino_u_format at build time; it travels with the pointer, the option
// 'wino4' may have changed since).  The option decides only whether the F(4x4) kernels run; the layout read is always the image's own.
static int run_wino(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int N, int H, int W, int Ci, int Co, int flip, int lrelu, hipStream_t s, const float* u_img = nullptr, int u_fmt = 0) {
    if (u_img && !mrdis_wino_u_fmt_valid(Ci, Co, 0, u_fmt)) return MRDIS_EINVAL;
    const bool f4 = mrdis_opt(MRDIS_OPT_WINO4) != 0 && mrdis_opt(MRDIS_OPT_WINO_U);
    if (u_img && u_fmt == 5) {       // <= 32 couts: the narrow F(4x4) form; declined -> the F(2x2) kernel for 32 couts (no image path)
        if (f4) {
            int rc = mrdis_run_wino4r(x, ldx, bias, y, ldy, N, H, W, Ci, Co, lrelu, s, u_img);
            if (rc == MRDIS_EUNSUPPORTED) rc = mrdis_run_wino4n(x, ldx, bias, y, ldy, N, H, W, Ci, Co, lrelu, s, u_img);
            if (rc != MRDIS_EUNSUPPORTED) return rc;
        }
        u_img = nullptr;
    }
    if (u_img && u_fmt == 4) {
        if (f4) {
            const int rc = mrdis_run_wino4(x, ldx, bias, y, ldy, N, H, W, Ci, Co, lrelu, s, u_img);
            if (rc != MRDIS_EUNSUPPORTED) return rc;
        }
        u_img += mrdis_wino4_image_floats(Ci, Co, 0);  // the 16-point image of the same filter follows the 36-point one
    }
    if (Co > 32 && mrdis_opt(MRDIS_OPT_WINO_PIPE)) {
        const int rc = mrdis_run_wino2(x, ldx, w, bias, y, ldy, N, H, W, Ci, Co, flip, lrelu, s, u_img);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    return mrdis_run_wino(x, ldx, w, bias, y, ldy, N, H, W, Ci, Co, flip, lrelu, s);
}
static bool wino_wanted(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad) {
    const int mode = (int)mrdis_opt(MRDIS_OPT_WINO);          // MRDIS_WINO at load; mrdis_set_option("wino", v) afterwards
    if (mode == 0 || kh != 3 || kw != 3 || stride != 1 || pad != 1) return false;
    if (mode == 2) return Ci >= 8 && Co >= 8;
    if (Ci < 16 || Co < 32) return false;
    const int cg = Co > 32 ? 2 : 1;
    const long long nblk = (long long)N * mrdis_cdiv((H + 1) / 2, 8) * mrdis_cdiv((W + 1) / 2, 8) * mrdis_cdiv(Co, 32 * cg);
    // measured (tools/layer_bench.py, B = 32, pipelined kernel): 16x16 maps with 128 blocks still win (ana.up_4 105 -> 92 us, sp2.gamma+beta
    // 58 -> 50 us); 8x8 maps fill a quarter of a 16x16-output block and lose (sp1: 36 -> 47 us), 64 blocks lose (66 -> 87 us)
    if (cg == 2 && nblk >= 128 && H >= 16 && W >= 16) return true;
    return nblk >= (cg == 2 ? 256 : 512);
}

static bool bf16m_wanted(int dtype, const void* w_bf16, int Cred, int Cout) {
    return (dtype == MRDIS_DT_F32_BF16M || dtype == MRDIS_DT_BF16) && w_bf16 != nullptr && Cred % 16 == 0 && Cout % 4 == 0 && Cout >= 16;
}

// SPADE block, fused: gamma | beta convolution + InstanceNorm modulation in one launch.  fp32 (MRDIS_DT_F32): the SPADE epilogue of the
// pipelined Winograd kernel (mrdis_wino2.hip), only where that kernel is the kernel of choice for the 2C-cout layer; bf16 activations
// (MRDIS_DT_BF16): the SPADE epilogue of the pipelined bf16 kernel (mrdis_bf16p.hip).  MRDIS_EUNSUPPORTED otherwise (the caller then runs
// mrdis_conv2d_fwd + mrdis_instnorm_spade_fwd).
int mrdis_run_wino2_spade(const float* x, int ldx, const float* w, const float* bias, const float* z, int ldz, const float* mean, const float* rstd,
                          float* mix, int ldmix, float* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s, const float* u_img);
int mrdis_run_bconv3_spade(const void* x, int ldx, const void* w_bf16, const float* bias, const void* z, int ldz, const float* mean, const float* rstd,
                           void* mix, int ldmix, void* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s);
int mrdis_run_bconv4_spade(const void* x, int ldx, const void* w_bf16, const float* bias, const void* z, int ldz, const float* mean, const float* rstd,
                           void* mix, int ldmix, void* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s);
extern "C" int mrdis_conv2d_fwd_spade(const void* x, int ldx, const float* w_tck, const void* w_bf16_tkc, const float* bias, const void* z, int ldz,
                                      const float* mean, const float* rstd, void* mix, int ldmix, void* gamma, int ldg,
                                      int N, int H, int W, int Ci, int C, int dtype, const float* w_wino, int w_wino_fmt, void* stream) {
    if (!x || !bias || !z || !mean || !rstd || !mix || !gamma || N < 1 || H < 1 || W < 1 || Ci < 1 || C < 1) return MRDIS_EINVAL;
    if (dtype == MRDIS_DT_BF16) {
        if (!w_bf16_tkc) return MRDIS_EINVAL;
        const int rc4 = mrdis_run_bconv4_spade(x, ldx, w_bf16_tkc, bias, z, ldz, mean, rstd, mix, ldmix, gamma, ldg, N, H, W, Ci, C, (hipStream_t)stream);
        if (rc4 != MRDIS_EUNSUPPORTED) return rc4;
        return mrdis_run_bconv3_spade(x, ldx, w_bf16_tkc, bias, z, ldz, mean, rstd, mix, ldmix, gamma, ldg, N, H, W, Ci, C, (hipStream_t)stream);
    }
    if (dtype != MRDIS_DT_F32 || !w_tck) return dtype == MRDIS_DT_F32 ? MRDIS_EINVAL : MRDIS_EUNSUPPORTED;
    if (!mrdis_opt(MRDIS_OPT_WINO_PIPE) || !wino_wanted(N, H, W, Ci, 2 * C, 3, 3, 1, 1)) return MRDIS_EUNSUPPORTED;
    if (w_wino && !mrdis_wino_u_fmt_valid(Ci, 2 * C, C, w_wino_fmt)) return MRDIS_EINVAL;
    if (w_wino && w_wino_fmt == 5) return MRDIS_EINVAL;           // a fused gamma | beta image is never the narrow form
    if (w_wino && w_wino_fmt == 4) {          // the image is the F(4x4) one (mrdis_wino4.hip), followed by the 16-point one
        if (mrdis_opt(MRDIS_OPT_WINO_U) && mrdis_opt(MRDIS_OPT_WINO4)) {
            const int rc4 = mrdis_run_wino4_spade((const float*)x, ldx, bias, (const float*)z, ldz, mean, rstd, (float*)mix, ldmix, (float*)gamma, ldg, N, H, W, Ci, C,
                                                  (hipStream_t)stream, w_wino);
            if (rc4 != MRDIS_EUNSUPPORTED) return rc4;
        }
        w_wino += mrdis_wino4_image_floats(Ci, 2 * C, C);
    }
    return mrdis_run_wino2_spade((const float*)x, ldx, w_tck, bias, (const float*)z, ldz, mean, rstd, (float*)mix, ldmix, (float*)gamma, ldg, N, H, W, Ci, C,
                                 (hipStream_t)stream, w_wino);
}

// mrdis_pointwise.hip: the 1x1 decoder head (16 -> <= 8 channels) as streaming kernels
int mrdis_run_pw_fwd(const void* x, int ldx, const float* w_tck, const float* bias, float* y, int ldy, long long npix, int Ci, int Co, int lrelu, int x16_bf16, hipStream_t s);
// mrdis_wgrad_s2.hip: forward of the stride-2 first layers (Cin <= 7)
int mrdis_run_conv_s2_fwd(const float* x, int ldx, const float* w_tck, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co,
                          int kh, int kw, int stride, int pad, int lrelu, hipStream_t s);
int mrdis_run_dgrad_s2(const float* dy, int lddy, const float* w_tkc, float* dx, int lddx, int N, int H, int W, int Ci, int Co,
                       int kh, int kw, int stride, int pad, hipStream_t s);
// mrdis_co4.hip: 3x3 s1 p1 with 4 output channels: window-free GEMM + tap gather
int mrdis_run_co4(const void* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int flip, int lrelu, hipStream_t s,
                  int x_bf16 = 0, int wld = 4);
// mrdis_c16.hip: 3x3 s1 p1 with 16 output channels, filter in registers
int mrdis_run_c16(const float* x, int ldx, const float* w_tck, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu, hipStream_t s);
int mrdis_run_c16t_split6(const float* x, int ldx, const float* w_t_ci_co, float* y, int ldy, int N, int H, int W, int flip, hipStream_t s);      // mrdis_c16.hip: 16 -> 32, six bf16 products
int mrdis_run_pw_dgrad(const float* dy, int lddy, const float* w_tkc, void* dx, int lddx, long long npix, int Ci, int Co, int x16_bf16, hipStream_t s);

extern "C" int mrdis_conv2d_fwd(const void* x_, int ldx, const float* w_tck, const void* w_bf16_tkc, const float* bias,
                                void* y_, int ldy, int N, int H, int W, int Ci, int Co,
                                int kh, int kw, int stride, int pad, int epilogue, int dtype, const float* w_wino, int w_wino_fmt, void* stream) {
    if (dtype < MRDIS_DT_F32 || dtype > MRDIS_DT_XF32_YBF16) return MRDIS_EUNSUPPORTED;
    const void* s6_img = nullptr;                  // w_wino_fmt 6: not a Winograd image but the six-product image of w_tck (mrdis_s6_filter_image): the layer goes to mrdis_s6conv.hip
    if (w_wino && w_wino_fmt == MRDIS_S6_IMAGE_FMT) { s6_img = dtype == MRDIS_DT_F32 ? w_wino : nullptr; w_wino = nullptr; w_wino_fmt = 0; }
    const float* x = reinterpret_cast<const float*>(x_); float* y = reinterpret_cast<float*>(y_);   // bf16 views when dtype == MRDIS_DT_BF16
    const bool st_bf16 = dtype == MRDIS_DT_BF16;
    int Ho, Wo;
    int rc = check_conv_geom(N, H, W, Ci, Co, kh, kw, stride, pad, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w_tck || !y || ldx < Ci || ldy < Co) return MRDIS_EINVAL;
    if (dtype == MRDIS_DT_XBF16_YF32) {       // mixed storage, bf16 in / fp32 out: the 1x1 decoder head (16 -> <= 8); the 3x3 C -> 4 layer (ana_dec.output)
        if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && Co == 4)       // w_tck is the [9][Ci][16] layout (columns >= 4 zero), bias 4 (or 16) floats
            return mrdis_run_co4(x_, ldx, w_tck, bias, y, ldy, N, H, W, Ci, Co, 0, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, (hipStream_t)stream, 1, 16);
        if (!(kh == 1 && kw == 1 && stride == 1 && pad == 0)) return MRDIS_EUNSUPPORTED;
        return mrdis_run_pw_fwd(x_, ldx, w_tck, bias, y, ldy, (long long)N * H * W, Ci, Co, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, 1, (hipStream_t)stream);
    }
    if (dtype == MRDIS_DT_XF32_YBF16) {       // mixed storage, fp32 in / bf16 out: the 4 -> C si_layers; w_tck is [9][16][Co] (rows >= 4 zero)
        if (!c4_eligible(x, ldx, ldy, N, H, W, Ci, Co, kh, kw, stride, pad, 2) || Co % 2 != 0 || ldy % 2 != 0 || (((uintptr_t)y_) & 3) != 0 ||
            (epilogue & MRDIS_EPI_LRELU) || mrdis_opt(MRDIS_OPT_NOC4)) return MRDIS_EUNSUPPORTED;
        return run_c4conv(x, ldx, w_tck, bias, y, ldy, N, H, W, Co, epilogue, (hipStream_t)stream, 0, 16, true);
    }
    if (!st_bf16 && kh == 1 && kw == 1 && stride == 1 && pad == 0) {
        rc = mrdis_run_pw_fwd(x, ldx, w_tck, bias, y, ldy, (long long)N * H * W, Ci, Co, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (!st_bf16 && c4_eligible(x, ldx, ldy, N, H, W, Ci, Co, kh, kw, stride, pad) && !mrdis_opt(MRDIS_OPT_NOC4))
        return run_c4conv(x, ldx, w_tck, bias, y, ldy, N, H, W, Co, epilogue, (hipStream_t)stream);
    const bool bf = bf16m_wanted(dtype, w_bf16_tkc, Ci, Co);
    if (st_bf16 && !bf) return MRDIS_EUNSUPPORTED;                // bf16 views: only the bf16 kernels may touch them
    if (!st_bf16 && stride == 2 && Ci <= 7) {
        rc = mrdis_run_conv_s2_fwd(x, ldx, w_tck, bias, y, ldy, N, H, W, Ci, Co, kh, kw, stride, pad, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (!st_bf16 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Co == 4) {
        rc = mrdis_run_co4(x, ldx, w_tck, bias, y, ldy, N, H, W, Ci, Co, 0, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (!bf && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Co == 16) {
        rc = mrdis_run_c16(x, ldx, w_tck, bias, y, ldy, N, H, W, Ci, Co, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (!bf && !s6_img && wino_wanted(N, H, W, Ci, Co, kh, kw, stride, pad)) {
        rc = run_wino(x, ldx, w_tck, bias, y, ldy, N, H, W, Ci, Co, 0, (epilogue & MRDIS_EPI_LRELU) ? 1 : 0, (hipStream_t)stream, w_wino, w_wino_fmt);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    TapConvParams p{};
    p.in = x; p.w = w_tck; p.bias = bias; p.out = y;
    p.N = N; p.Hin = H; p.Win = W; p.Cin = Ci; p.ldin = ldx;
    p.Hout = Ho; p.Wout = Wo; p.Cout = Co; p.ldout = ldy;
    p.A = Ho; p.B = Wo; p.os = 1; p.oh0 = 0; p.ow0 = 0; p.is = stride;
    p.ntaps = kh * kw;
    for (int r = 0; r < kh; ++r)
        for (int s_ = 0; s_ < kw; ++s_) {
            const int t = r * kw + s_;
            p.dh[t] = r - pad; p.dw[t] = s_ - pad; p.widx[t] = t;
        }
    p.epilogue = epilogue;
    p.w_bf16 = bf ? w_bf16_tkc : nullptr; p.dtype = dtype;
    p.s6_img = s6_img; p.s6_taps = kh * kw;
    return run_tapconv(p, (hipStream_t)stream);
}

extern "C" int mrdis_conv2d_bwd_data(const void* dy_, int lddy, const float* w_tkc, const void* w_bf16_tck,
                                     void* dx_, int lddx, int N, int H, int W, int Ci, int Co,
                                     int kh, int kw, int stride, int pad, int dtype, const float* w_wino, int w_wino_fmt, void* stream) {
    if (dtype != MRDIS_DT_F32 && dtype != MRDIS_DT_F32_BF16M && dtype != MRDIS_DT_BF16 && dtype != MRDIS_DT_XBF16_YF32 && dtype != MRDIS_DT_XF32_YBF16) return MRDIS_EUNSUPPORTED;
    const void* s6_img = nullptr;                  // w_wino_fmt 6: the six-product image of w_tkc (reduction over Co, Ci output channels)
    if (w_wino && w_wino_fmt == MRDIS_S6_IMAGE_FMT) { s6_img = dtype == MRDIS_DT_F32 ? w_wino : nullptr; w_wino = nullptr; w_wino_fmt = 0; }
    const float* dy = reinterpret_cast<const float*>(dy_); float* dx = reinterpret_cast<float*>(dx_);
    const bool st_bf16 = dtype == MRDIS_DT_BF16;
    int Ho, Wo;
    int rc = check_conv_geom(N, H, W, Ci, Co, kh, kw, stride, pad, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !w_tkc || !dx || lddy < Co || lddx < Ci) return MRDIS_EINVAL;
    if (dtype == MRDIS_DT_XF32_YBF16) {       // bf16 storage, dy bf16 -> dx fp32: the data gradient of the 4 -> C si_layers (a C -> 4 convolution of dy with the
        // taps reversed); w_tkc is the [9][Co][16] layout (columns >= 4 zero)
        if (!(kh == 3 && kw == 3 && stride == 1 && pad == 1 && Ci == 4) || !dy_ || !w_tkc || !dx_) return MRDIS_EUNSUPPORTED;
        return mrdis_run_co4(dy_, lddy, w_tkc, nullptr, dx, lddx, N, H, W, Co, Ci, 1, 0, (hipStream_t)stream, 1, 16);
    }
    if (dtype == MRDIS_DT_XBF16_YF32) {       // bf16 storage, dy fp32 -> dx bf16: the 1x1 head (<= 8 -> 16 channels) and the C <- 4 layer (ana_dec.output)
        if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && Co == 4) {
            // dx of a Ci -> 4 layer: a 4 -> Ci convolution of the fp32 dy with the taps reversed, bf16 out; w_tkc is [9][16][Ci] (rows >= 4 zero: the
            // 16-row layout the padded bf16 kernels use for the same layer)
            if (!c4_eligible(dy, lddy, lddx, N, H, W, Co, Ci, kh, kw, stride, pad, 2) || Ci % 2 != 0 || lddx % 2 != 0 || (((uintptr_t)dx_) & 3) != 0 ||
                mrdis_opt(MRDIS_OPT_NOC4)) return MRDIS_EUNSUPPORTED;
            return run_c4conv(dy, lddy, w_tkc, nullptr, dx, lddx, N, H, W, Ci, 0, (hipStream_t)stream, 1, 16, true);
        }
        if (!(kh == 1 && kw == 1 && stride == 1 && pad == 0)) return MRDIS_EUNSUPPORTED;
        return mrdis_run_pw_dgrad(dy, lddy, w_tkc, dx_, lddx, (long long)N * H * W, Ci, Co, 1, (hipStream_t)stream);
    }
    if (!st_bf16 && kh == 1 && kw == 1 && stride == 1 && pad == 0) {
        rc = mrdis_run_pw_dgrad(dy, lddy, w_tkc, dx, lddx, (long long)N * H * W, Ci, Co, 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    TapConvParams base{};
    base.in = dy; base.w = w_tkc; base.bias = nullptr; base.out = dx;
    base.N = N; base.Hin = Ho; base.Win = Wo; base.Cin = Co; base.ldin = lddy;
    base.Hout = H; base.Wout = W; base.Cout = Ci; base.ldout = lddx;
    base.is = 1; base.epilogue = 0;
    base.s6_img = s6_img; base.s6_taps = kh * kw;
    const bool bf = bf16m_wanted(dtype, w_bf16_tck, Co, Ci);      // the data gradient reduces over Co and produces Ci channels
    if (st_bf16 && !bf) return MRDIS_EUNSUPPORTED;
    base.w_bf16 = bf ? w_bf16_tck : nullptr; base.dtype = dtype;
    if (!st_bf16 && stride == 2 && Ci <= 7) {
        rc = mrdis_run_dgrad_s2(dy, lddy, w_tkc, dx, lddx, N, H, W, Ci, Co, kh, kw, stride, pad, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (!st_bf16 && stride == 1 && kh == 3 && kw == 3 && pad == 1 && Ci == 4) {
        // dx of a 4 -> Co layer (the SPADE si_layers): a Co -> 4 convolution of dy with the taps reversed; [tap][Co][4] is the filter layout
        rc = mrdis_run_co4(dy, lddy, w_tkc, nullptr, dx, lddx, N, H, W, Co, Ci, 1, 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (stride == 1) {
        // dx of a Ci <- 4 layer (ana_dec.output): a 4 -> Ci convolution of dy with the taps reversed; [tap][Co=4][Ci] is
        // exactly the [tap][4][Cout'] filter layout of the Cin = 4 kernel
        if (!st_bf16 && c4_eligible(dy, lddy, lddx, N, H, W, Co, Ci, kh, kw, stride, pad) && !mrdis_opt(MRDIS_OPT_NOC4))
            return run_c4conv(dy, lddy, w_tkc, nullptr, dx, lddx, N, H, W, Ci, 0, (hipStream_t)stream, 1);
        if (!bf && !st_bf16 && kh == 3 && kw == 3 && pad == 1 && Co == 16 && Ci == 32) {      // sp6.out's data gradient on the six-product kernel (option split6; mrdis_c16.hip)
            rc = mrdis_run_c16t_split6(dy, lddy, w_tkc, dx, lddx, N, H, W, 1, (hipStream_t)stream);
            if (rc != MRDIS_EUNSUPPORTED) return rc;
        }
        if (!bf && !s6_img && wino_wanted(N, H, W, Co, Ci, kh, kw, stride, pad)) {
            rc = run_wino(dy, lddy, w_tkc, nullptr, dx, lddx, N, H, W, Co, Ci, 1, 0, (hipStream_t)stream, w_wino, w_wino_fmt);
            if (rc != MRDIS_EUNSUPPORTED) return rc;
        }
        TapConvParams p = base;
        p.A = H; p.B = W; p.os = 1; p.oh0 = 0; p.ow0 = 0;
        p.ntaps = kh * kw;
        for (int r = 0; r < kh; ++r)
            for (int s_ = 0; s_ < kw; ++s_) {
                const int t = r * kw + s_;
                p.dh[t] = pad - r; p.dw[t] = pad - s_; p.widx[t] = t;
            }
        return run_tapconv(p, (hipStream_t)stream);
    }
    // stride 2: dx[hi] gathers dy[(hi + pad - r)/2] for the taps r with (hi + pad - r) even.
    TapLaunch L[4];
    const bool pack = !mrdis_opt(MRDIS_OPT_NOW16) && !mrdis_opt(MRDIS_OPT_NOPACK);      // (debug_now16 / debug_nopack = 1: four launches, as before)
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            TapConvParams p = base;
            p.A = (H - ph + 1) / 2; p.B = (W - pw + 1) / 2;
            p.os = 2; p.oh0 = ph; p.ow0 = pw;
            p.ntaps = 0;
            for (int r = 0; r < kh; ++r) {
                if (((ph + pad - r) & 1) != 0) continue;
                for (int s_ = 0; s_ < kw; ++s_) {
                    if (((pw + pad - s_) & 1) != 0) continue;
                    const int t = p.ntaps++;
                    // exact: numerator is even; arithmetic shift keeps floor semantics for negatives
                    p.dh[t] = (ph + pad - r) >> 1; p.dw[t] = (pw + pad - s_) >> 1; p.widx[t] = r * kw + s_;
                }
            }
            if (p.ntaps == 0) return MRDIS_EUNSUPPORTED;   // would need a zero fill; not on the path
            rc = run_tapconv(p, (hipStream_t)stream, pack ? &L[2 * ph + pw] : nullptr);
            if (rc) return rc;
        }
    if (pack) {
        {   // classes planned for the bf16 kernel
            bool any = false;
            for (int k = 0; k < 4; ++k) any = any || L[k].b.set;
            if (any) {
                BConvLaunch LB[4];
                for (int k = 0; k < 4; ++k) LB[k] = L[k].b;
                rc = mrdis_launch_bconv_planned(LB, (hipStream_t)stream);
                if (rc) return rc;
            }
        }
        {   // classes planned for the six-product kernel
            bool any = false;
            for (int k = 0; k < 4; ++k) any = any || L[k].s6.set;
            if (any) {
                S6ConvLaunch LS[4];
                for (int k = 0; k < 4; ++k) LS[k] = L[k].s6;
                rc = mrdis_launch_s6conv_planned(LS, (hipStream_t)stream);
                if (rc) return rc;
            }
        }
        // classes that were planned for tapconv_kernel (not launched yet): one launch if they agree on the instantiation, else one each
        bool all = true;
        for (int k = 0; k < 4; ++k) all = all && L[k].set;
        bool same = all;
        for (int k = 1; k < 4 && same; ++k)
            same = L[k].KC == L[0].KC && L[k].BN == L[0].BN && L[k].BM == L[0].BM && L[k].p.prefetch == L[0].p.prefetch;
        if (same) {
#define TCP_CASE(kc, bn) if (L[0].KC == kc && L[0].BN == bn) return launch_tapconv_pack_t<kc, bn>(L, (hipStream_t)stream)
            TCP_CASE(4, 32); TCP_CASE(4, 64); TCP_CASE(8, 32); TCP_CASE(8, 64); TCP_CASE(16, 32); TCP_CASE(16, 64);
#undef TCP_CASE
            return MRDIS_EUNSUPPORTED;
        }
        for (int k = 0; k < 4; ++k) {
            if (!L[k].set) continue;
#define TC1_CASE(kc, bn) if (L[k].KC == kc && L[k].BN == bn) rc = launch_tapconv_t<kc, bn>(L[k].p, L[k].lds, L[k].nblk, L[k].BM, (hipStream_t)stream)
            TC1_CASE(4, 32); TC1_CASE(4, 64); TC1_CASE(8, 32); TC1_CASE(8, 64); TC1_CASE(16, 32); TC1_CASE(16, 64);
#undef TC1_CASE
            if (rc) return rc;
        }
    }
    return MRDIS_OK;
}

// =========================================================================== weight gradient
#define WG_MAX_SLOTS 32
#define WG_MAX_GROUPS 4
struct WgradParams {
    const float* x; const float* dy; float* slab;
    int N, Hin, Win, Ci, ldx;
    int A, B, Co, lddy;           // dy extents (Ho, Wo)
    int is, ntaps;                // ntaps = tap SLOTS (nG * J * TPS); slot_ok marks the real ones
    int dh[WG_MAX_SLOTS], dw[WG_MAX_SLOTS];
    unsigned slot_ok;
    // per tap group g: the input view it reads (stride-2 layers: one parity class per group)
    long long g_off[WG_MAX_GROUPS];
    int g_Hin[WG_MAX_GROUPS], g_Win[WG_MAX_GROUPS], g_dhmin[WG_MAX_GROUPS], g_dwmin[WG_MAX_GROUPS];
    int NB, TH, TW, TinH, TinW, tilesA, tilesB, tilesN, numTiles;
    int CW, TPS;                  // channels per M sub-tile row group (<=32), taps per sub-tile
    int nCi, nCo, nG, base;       // ci chunks (32), co chunks (32), tap groups, base = nCi*nCo*nG
    int splits;
    int vec_x, vec_dy;
    float* bias_slab;             // [splits][nCo*32] column sums of dy, or nullptr
    long long x_img; int x_row, x_pix;   // element pitches of the input view (image, row, pixel)
};

#define WG_TAB_INTS 192   // tab_in[128] tap_xoff[16] + pad

template <int J>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab_in = reinterpret_cast<int*>(smem);
    int* tab_pos = tab_in + 128;          // packed (n,a,b) validity -> dy pixel index or -1 (per tile, rebuilt)
    float* dys = smem + 256;              // [128][32]
    float* xs = dys + 128 * 32;           // [npix_in][S]
    const int S = p.CW + 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, e = lane & 31;
    const int bid = blockIdx.x;
    const int split = bid / p.base;
    int b = bid - split * p.base;
    const int coc = b % p.nCo; b /= p.nCo;
    const int cic = b % p.nCi;
    const int g = b / p.nCi;
    const float* __restrict__ xg = p.x + p.g_off[g];
    const int gHin = p.g_Hin[g], gWin = p.g_Win[g], g_dh_min = p.g_dhmin[g], g_dw_min = p.g_dwmin[g];
    const int c_lo = cic * 32, co_lo = coc * 32;

    const int tinHW = p.TinH * p.TinW, npix_in = p.NB * tinHW, npos = p.NB * p.TH * p.TW;
    if (tid < 128) {
        const int m = tid;
        int tin = 0;
        if (m < npos) {
            const int nb = m / (p.TH * p.TW);
            const int rem = m - nb * p.TH * p.TW;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = ((nb * p.TinH + ty * p.is) * p.TinW + tx * p.is) * S;
        }
        tab_in[m] = tin;
    }
    // per-lane sub-tile row -> (tap, ci) -> offset inside the staged input tile
    int loff[J];
    unsigned lvalid = 0;
    {
        const int tl = e / p.CW, cl = e - tl * p.CW;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int tap = (g * J + j) * p.TPS + tl;
            const bool ok = tap < p.ntaps && ((p.slot_ok >> tap) & 1u) && (c_lo + cl) < p.Ci;
            loff[j] = ok ? ((p.dh[tap] - g_dh_min) * p.TinW + (p.dw[tap] - g_dw_min)) * S + cl : 0;
            lvalid |= (ok ? 1u : 0u) << j;
        }
    }
    __syncthreads();
    // positions handled by this wave: pairs q = wave + 4*pp, position m = 2q + half

    f32x16 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const bool do_bias = (p.bias_slab != nullptr) && cic == 0 && g == 0;
    float bsum = 0.f;

    for (int tile = split; tile < p.numTiles; tile += p.splits) {
        int tt = tile;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA;
        const int tn = tt / p.tilesA;
        const int a0 = ta * p.TH, b0 = tb * p.TW, n0 = tn * p.NB;
        const int h_org = a0 * p.is + g_dh_min, w_org = b0 * p.is + g_dw_min;
        __syncthreads();
        if (tid < 128) {
            const int m = tid;
            int pos = -1;
            if (m < npos) {
                const int nb = m / (p.TH * p.TW);
                const int rem = m - nb * p.TH * p.TW;
                const int ty = rem / p.TW, tx = rem - ty * p.TW;
                const int n = n0 + nb, a = a0 + ty, bb = b0 + tx;
                if (n < p.N && a < p.A && bb < p.B) pos = (n * p.A + a) * p.B + bb;
            }
            tab_pos[m] = pos;
        }
        // stage x tile (channels c_lo .. c_lo+CW)
        if (p.vec_x) {
            const int Q = p.CW >> 2;
            for (int idx = tid; idx < npix_in * Q; idx += 256) {
                const int pi = idx / Q, q = idx - pi * Q;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix, c = c_lo + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.N && (unsigned)h < (unsigned)gHin && (unsigned)w_ < (unsigned)gWin && c < p.Ci)
                    v = *reinterpret_cast<const float4*>(xg + (long long)n * p.x_img + (long long)h * p.x_row + (long long)w_ * p.x_pix + c);
                float* d = xs + pi * S + 4 * q;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < npix_in * p.CW; idx += 256) {
                const int pi = idx / p.CW, k = idx - pi * p.CW;
                const int nb = pi / tinHW;
                const int rem = pi - nb * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int n = n0 + nb, h = h_org + iy, w_ = w_org + ix, c = c_lo + k;
                float v = 0.f;
                if (n < p.N && (unsigned)h < (unsigned)gHin && (unsigned)w_ < (unsigned)gWin && c < p.Ci)
                    v = xg[(long long)n * p.x_img + (long long)h * p.x_row + (long long)w_ * p.x_pix + c];
                xs[pi * S + k] = v;
            }
        }
        __syncthreads();   // tab_pos visible
        // stage dy tile [128][32]; invalid positions / couts are zero rows, which also
        // silences whatever the x tile holds there
        if (p.vec_dy) {
            for (int idx = tid; idx < 128 * 8; idx += 256) {
                const int m = idx >> 3, q = idx & 7;
                const int pos = tab_pos[m], co = co_lo + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pos >= 0 && co < p.Co)
                    v = *reinterpret_cast<const float4*>(p.dy + (long long)pos * p.lddy + co);
                *reinterpret_cast<float4*>(dys + m * 32 + 4 * q) = v;
            }
        } else {
            for (int idx = tid; idx < 128 * 32; idx += 256) {
                const int m = idx >> 5, q = idx & 31;
                const int pos = tab_pos[m], co = co_lo + q;
                float v = 0.f;
                if (pos >= 0 && co < p.Co) v = p.dy[(long long)pos * p.lddy + co];
                dys[idx] = v;
            }
        }
        __syncthreads();
        if (do_bias) {   // fused bias gradient: column sums of the staged dy tile (block-uniform branch)
            const int co_ = tid & 31, part = tid >> 5;
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += dys[(part * 16 + r) * 32 + co_];
        }
        {   // software-pipelined over the wave's 16 position pairs: operands of pair pp+1 (and the LDS
            // offset of pair pp+2) are fetched before the J MFMAs of pair pp issue.
            float an[J], bn;
            int ti_nn;
            {
                const int m0 = 2 * wave + half;
                const int ti0 = tab_in[m0];
                bn = dys[m0 * 32 + e];
#pragma unroll
                for (int j = 0; j < J; ++j) an[j] = xs[ti0 + loff[j]];
                ti_nn = tab_in[2 * (wave + 4) + half];
            }
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                float av[J];
                const float bv = bn;
#pragma unroll
                for (int j = 0; j < J; ++j) av[j] = ((lvalid >> j) & 1u) ? an[j] : 0.f;
                const int pn = pp + 1 < 16 ? pp + 1 : 15, pnn = pp + 2 < 16 ? pp + 2 : 15;
                const int m1 = 2 * (wave + 4 * pn) + half;
                bn = dys[m1 * 32 + e];
#pragma unroll
                for (int j = 0; j < J; ++j) an[j] = xs[ti_nn + loff[j]];
                ti_nn = tab_in[2 * (wave + 4 * pnn) + half];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < J; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv, acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // cross-wave reduction through LDS (fixed order), then slab[split][b][j][32][32]
    float* red = dys;                     // 4 waves x 1024 floats
    float* out = p.slab + (((long long)split * p.base + (bid - split * p.base)) * J) * 1024;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave * 1024 + row * 32 + e] = acc[j][r];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            out[j * 1024 + i] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
        }
    }
    if (do_bias) {
        __syncthreads();
        red[tid] = bsum;                  // [part(8)][co(32)]
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 32 + tid];
            p.bias_slab[((long long)split * p.nCo + coc) * 32 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------- LDS-DMA variant (Ci % 32 == 0, Co % 4 == 0)
// The plain kernel above is staging-bound: with 144+ accumulator registers only two workgroups fit
// per CU and nothing hides the global->register->LDS round trip of the next tile.  Here both tiles
// are rows of exactly 128 B ([pixel][32 ci] and [position][32 co]; for Ci >= 32 the A operand reads 32
// consecutive channels of one pixel, so no padding is needed), which is the shape
// `global_load_lds_dwordx4` wants: each wave-instruction lands 8 rows (1 KiB) straight in LDS with a
// per-lane source address, out-of-image pixels / positions are redirected to a zero page.  The
// tile after the current one is in flight (second LDS buffer, zero VGPRs) during the MFMAs; one barrier
// per tile; all index arithmetic (divisions) is hoisted out of the tile loop.
#define WGD_XSLOTS 7
template <int J>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradParams p, int XR /* staged x rows, multiple of 8 */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab_in = reinterpret_cast<int*>(smem);           // [128] position -> x row offset (floats)
    float* dys0 = smem + 128;                             // [2][128*32]
    float* xs0 = dys0 + 2 * 128 * 32;                     // [2][XR*32]

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, e = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int split = bid / p.base;
    int b = bid - split * p.base;
    const int coc = b % p.nCo; b /= p.nCo;
    const int cic = b % p.nCi;
    const int g = b / p.nCi;
    const float* __restrict__ xg = p.x + p.g_off[g];
    const int gHin = p.g_Hin[g], gWin = p.g_Win[g], g_dh_min = p.g_dhmin[g], g_dw_min = p.g_dwmin[g];
    const int c_lo = cic * 32, co_lo = coc * 32;
    const int tinHW = p.TinH * p.TinW, npix_in = p.NB * tinHW, npos = p.NB * p.TH * p.TW;
    const int thw = p.TH * p.TW;

    if (tid < 128) {
        int tin = 0;
        if (tid < npos) {
            const int nb = tid / thw, rem = tid - nb * thw, ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = ((nb * p.TinH + ty * p.is) * p.TinW + tx * p.is) * 32;
        }
        tab_in[tid] = tin;
    }
    // tile-invariant staging descriptors: which (image, row, col) of the tile this lane fetches per DMA slot
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
    int xd[WGD_XSLOTS], dd[4];
#pragma unroll
    for (int k = 0; k < WGD_XSLOTS; ++k) {
        const int pi = 8 * (wave + 4 * k) + r8;
        xd[k] = -1;
        if (pi < npix_in) {
            const int nb = pi / tinHW, rem = pi - nb * tinHW, iy = rem / p.TinW, ix = rem - iy * p.TinW;
            xd[k] = (nb << 20) | (iy << 10) | ix;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = 8 * (wave + 4 * k) + r8;
        dd[k] = -1;
        if (m < npos) {
            const int nb = m / thw, rem = m - nb * thw, ty = rem / p.TW, tx = rem - ty * p.TW;
            dd[k] = (nb << 20) | (ty << 10) | tx;
        }
    }
    int loff[J];
    unsigned lvalid = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int tap = g * J + j;
        const bool ok = tap < p.ntaps && ((p.slot_ok >> tap) & 1u);
        loff[j] = ok ? ((p.dh[tap] - g_dh_min) * p.TinW + (p.dw[tap] - g_dw_min)) * 32 + e : 0;
        lvalid |= (ok ? 1u : 0u) << j;
    }
    const float* zero = g_mrdis_zero_page + c4;
    const int xslots = XR >> 3;

    auto issue_tile = [&](int tile, int buf) {
        int tt = tile;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA;
        const int tn = tt / p.tilesA;
        const int a0 = ta * p.TH, b0 = tb * p.TW, n0 = tn * p.NB;
        const int h_org = a0 * p.is + g_dh_min, w_org = b0 * p.is + g_dw_min;
        float* xs = xs0 + buf * (XR * 32);
        float* dys = dys0 + buf * (128 * 32);
#pragma unroll
        for (int k = 0; k < WGD_XSLOTS; ++k) {
            const int q = wave + 4 * k;                 // wave-uniform slot
            if (q < xslots) {
                const float* src = zero;
                if (xd[k] >= 0) {
                    const int n = n0 + (xd[k] >> 20), h = h_org + ((xd[k] >> 10) & 1023), w_ = w_org + (xd[k] & 1023);
                    if (n < p.N && (unsigned)h < (unsigned)gHin && (unsigned)w_ < (unsigned)gWin)
                        src = xg + (long long)n * p.x_img + (long long)h * p.x_row + (long long)w_ * p.x_pix + c_lo + c4;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + q * 256), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = wave + 4 * k;
            const float* src = zero;
            if (dd[k] >= 0) {
                const int n = n0 + (dd[k] >> 20), a = a0 + ((dd[k] >> 10) & 1023), bb = b0 + (dd[k] & 1023);
                if (n < p.N && a < p.A && bb < p.B && co_lo + c4 < p.Co)      // couts beyond Co (Co % 4 == 0) read zeros
                    src = p.dy + ((long long)(n * p.A + a) * p.B + bb) * p.lddy + co_lo + c4;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dys + q * 256), 16, 0, 0);
        }
    };

    f32x16 acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const bool do_bias = (p.bias_slab != nullptr) && cic == 0 && g == 0;
    float bsum = 0.f;

    int tile = split, buf = 0;
    if (tile < p.numTiles) issue_tile(tile, 0);
    __syncthreads();                                     // (vmcnt(0) + barrier) first tile landed, tab_in visible
    for (; tile < p.numTiles; tile += p.splits) {
        if (tile + p.splits < p.numTiles) issue_tile(tile + p.splits, buf ^ 1);
        const float* xs = xs0 + buf * (XR * 32);
        const float* dys = dys0 + buf * (128 * 32);
        if (do_bias) {
            const int co_ = tid & 31, part = tid >> 5;
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += dys[(part * 16 + r) * 32 + co_];
        }
        {
            float an[J], bn;
            int ti_nn;
            {
                const int m0 = 2 * wave + half;
                const int ti0 = tab_in[m0];
                bn = dys[m0 * 32 + e];
#pragma unroll
                for (int j = 0; j < J; ++j) an[j] = xs[ti0 + loff[j]];
                ti_nn = tab_in[2 * (wave + 4) + half];
            }
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                float av[J];
                const float bv = bn;
#pragma unroll
                for (int j = 0; j < J; ++j) av[j] = ((lvalid >> j) & 1u) ? an[j] : 0.f;
                const int pn = pp + 1 < 16 ? pp + 1 : 15, pnn = pp + 2 < 16 ? pp + 2 : 15;
                const int m1 = 2 * (wave + 4 * pn) + half;
                bn = dys[m1 * 32 + e];
#pragma unroll
                for (int j = 0; j < J; ++j) an[j] = xs[ti_nn + loff[j]];
                ti_nn = tab_in[2 * (wave + 4 * pnn) + half];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < J; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv, acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                 // drains this wave's DMA (vmcnt(0)) and orders the buffers
        buf ^= 1;
    }
    float* red = dys0;
    float* out = p.slab + (((long long)split * p.base + (bid - split * p.base)) * J) * 1024;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            red[wave * 1024 + row * 32 + e] = acc[j][r];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            out[j * 1024 + i] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
        }
    }
    if (do_bias) {
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 32 + tid];
            p.bias_slab[((long long)split * p.nCo + coc) * 32 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------- thin variant (Co == 4 or Ci == 4)
// ana_dec.output (64 -> 4) and every `si_layers` (4 -> C) have one side of the weight-gradient product only 4
// wide: a 32x32 MFMA tile is then 75-88 % padding and the kernels above run at 13-28 TF/s while the layer is
// really a streaming reduction (its floor is the HBM read of x and dy).  Here the narrow side stays a
// per-thread register vector and the product is packed FMAs: thread = (wide channel e, position slot), 8
// slots x 16 positions per 128-position tile, acc[tap][narrow] += wide[e] * narrow[k], the narrow operand
// read as an LDS broadcast; 3x3 layers keep the window in registers while a thread walks its 16 positions
// along a tile row (3 LDS reads per position instead of 9).  Tiles arrive by `global_load_lds` into a
// double buffer as in wgrad_dma_kernel, all index arithmetic hoisted out of the tile loop (a register-staged
// version of this kernel spent more time computing addresses than multiplying and lost to the MFMA kernels).
// Same tap groups and slab layout as the other kernels, so the fixed-order reduce is shared.
//   NARROW_X = false:  wide = x channels (chunk of 32, Ci % 32 == 0), narrow = 4 couts   (dW[t][ci=e][co=k])
//   NARROW_X = true :  wide = couts (chunk of 32), narrow = the 4 channels of x            (dW[t][ci=k][co=e])
template <int NT, int NN, bool NARROW_X>
__global__ __launch_bounds__(256) void wgrad_thin_dma_kernel(const WgradParams p, int J, int XR /* staged x rows */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int XW = NARROW_X ? 4 : 32;                 // floats per staged x row
    constexpr int RPI = NARROW_X ? 64 : 8;                // x rows one DMA wave-instruction lands
    constexpr int DYW = NARROW_X ? 32 : 4;                // floats per staged dy row (narrow dy: the 4 real couts only)
    constexpr int RPD = NARROW_X ? 8 : 64;                // dy rows per DMA wave-instruction
    constexpr int DSLOTS = 128 / RPD / 4 > 0 ? 128 / RPD / 4 : 1;   // dy DMA instructions per wave
    int* tab_in = reinterpret_cast<int*>(smem);           // [128] position -> x row offset (floats)
    float* dys0 = smem + 128;                             // [2][128*DYW]
    float* xs0 = dys0 + 2 * 128 * DYW;                    // [2][XR*XW]

    const int tid = threadIdx.x, lane = tid & 63, e = tid & 31, slot8 = tid >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int split = bid / p.base;
    int b = bid - split * p.base;
    const int coc = b % p.nCo; b /= p.nCo;
    const int cic = b % p.nCi;
    const int g = b / p.nCi;
    const float* __restrict__ xg = p.x + p.g_off[g];
    const int gHin = p.g_Hin[g], gWin = p.g_Win[g], g_dh_min = p.g_dhmin[g], g_dw_min = p.g_dwmin[g];
    const int c_lo = cic * 32, co_lo = coc * 32;
    const int tinHW = p.TinH * p.TinW, npix_in = p.NB * tinHW, npos = p.NB * p.TH * p.TW;
    const int thw = p.TH * p.TW;
    const int slot0 = g * J * p.TPS;

    if (tid < 128) {
        int tin = 0;
        if (tid < npos) {
            const int nb = tid / thw, rem = tid - nb * thw, ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = ((nb * p.TinH + ty * p.is) * p.TinW + tx * p.is) * XW;
        }
        tab_in[tid] = tin;
    }
    // tile-invariant staging descriptors
    const int xrow_l = NARROW_X ? lane : (lane >> 3);     // x row inside a DMA instruction
    const int xc4 = NARROW_X ? 0 : (lane & 7) * 4;        // float offset inside the row
    const int r8 = NARROW_X ? (lane >> 3) : lane, c4 = NARROW_X ? (lane & 7) * 4 : 0;   // dy row / float offset inside a DMA instruction
    int xd[WGD_XSLOTS], dd[DSLOTS];
#pragma unroll
    for (int k = 0; k < WGD_XSLOTS; ++k) {
        const int pi = RPI * (wave + 4 * k) + xrow_l;
        xd[k] = -1;
        if (pi < npix_in) {
            const int nb = pi / tinHW, rem = pi - nb * tinHW, iy = rem / p.TinW, ix = rem - iy * p.TinW;
            xd[k] = (nb << 20) | (iy << 10) | ix;
        }
    }
#pragma unroll
    for (int k = 0; k < DSLOTS; ++k) {
        const int m = RPD * (wave + 4 * k) + r8;
        dd[k] = -1;
        if (m < npos) {
            const int nb = m / thw, rem = m - nb * thw, ty = rem / p.TW, tx = rem - ty * p.TW;
            dd[k] = (nb << 20) | (ty << 10) | tx;
        }
    }
    int toff[NT];
    unsigned tvalid = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int sl = slot0 + t;
        const bool ok = t < J * p.TPS && sl < p.ntaps && ((p.slot_ok >> sl) & 1u);
        toff[t] = ok ? ((p.dh[sl] - g_dh_min) * p.TinW + (p.dw[sl] - g_dw_min)) * XW : 0;
        tvalid |= (ok ? 1u : 0u) << t;
    }
    const float* zero = g_mrdis_zero_page + (NARROW_X ? 0 : c4);
    const float* zero_dy = g_mrdis_zero_page + c4;
    const int xslots = XR / RPI;

    auto issue_tile = [&](int tile, int buf) {
        int tt = tile;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA;
        const int tn = tt / p.tilesA;
        const int a0 = ta * p.TH, b0 = tb * p.TW, n0 = tn * p.NB;
        const int h_org = a0 * p.is + g_dh_min, w_org = b0 * p.is + g_dw_min;
        float* xs = xs0 + buf * (XR * XW);
        float* dys = dys0 + buf * (128 * DYW);
#pragma unroll
        for (int k = 0; k < WGD_XSLOTS; ++k) {
            const int q = wave + 4 * k;                 // wave-uniform slot
            if (q < xslots) {
                const float* src = zero;
                if (xd[k] >= 0) {
                    const int n = n0 + (xd[k] >> 20), h = h_org + ((xd[k] >> 10) & 1023), w_ = w_org + (xd[k] & 1023);
                    if (n < p.N && (unsigned)h < (unsigned)gHin && (unsigned)w_ < (unsigned)gWin)
                        src = xg + (long long)n * p.x_img + (long long)h * p.x_row + (long long)w_ * p.x_pix + (NARROW_X ? 0 : c_lo + xc4);
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs + q * 256), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < DSLOTS; ++k) {
            const int q = wave + 4 * k;
            if (q * RPD >= 128) continue;               // wave-uniform
            const float* src = zero_dy;
            if (dd[k] >= 0) {
                const int n = n0 + (dd[k] >> 20), a = a0 + ((dd[k] >> 10) & 1023), bb = b0 + (dd[k] & 1023);
                if (n < p.N && a < p.A && bb < p.B && co_lo + c4 < p.Co)
                    src = p.dy + ((long long)(n * p.A + a) * p.B + bb) * p.lddy + co_lo + c4;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dys + q * 256), 16, 0, 0);
        }
    };

    const bool wide_ok = NARROW_X ? (co_lo + e < p.Co) : (c_lo + e < p.Ci);
    // the tap slots of a 3x3 stride-1 layer are r*3+c with unit offsets: the sliding-window loop applies
    // (a thread's 16 positions slot8*16 .. +15 never straddle a tile row when TW is a multiple of 16)
    const bool slide = NT == 9 && p.TW % 16 == 0 && npos == 128 && p.TinW == p.TW + 2 && p.TinH == p.TH + 2 && p.is == 1 && p.nG == 1 && tvalid == 0x1ffu;
    float acc[NT][NN];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < NN; ++k) acc[t][k] = 0.f;
    const bool do_bias = (p.bias_slab != nullptr) && cic == 0 && g == 0;
    float bsum = 0.f;

    int tile = split, buf = 0;
    if (tile < p.numTiles) issue_tile(tile, 0);
    __syncthreads();                                     // (vmcnt(0) + barrier) first tile landed, tab_in visible
    for (; tile < p.numTiles; tile += p.splits) {
        if (tile + p.splits < p.numTiles) issue_tile(tile + p.splits, buf ^ 1);
        const float* xs = xs0 + buf * (XR * XW);
        const float* dys = dys0 + buf * (128 * DYW);
        if (do_bias && e < DYW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += dys[(slot8 * 16 + r) * DYW + e];
        }
        if (NT == 9 && slide) {
            // 3x3 taps, tile rows of 16 positions: thread (e, slot8) walks its row left to right and keeps
            // the 3x3 window in registers -- one new window column (3 LDS reads) per position instead of 9
            const int rowbase = tab_in[slot8 * 16];
            if (NARROW_X) {
                float4 col[3][3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 3; ++r) col[c][r] = *reinterpret_cast<const float4*>(xs + rowbase + (r * p.TinW + c) * XW);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float wv = dys[(slot8 * 16 + i) * DYW + e];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float4 q = col[(i + c) % 3][r];
                            acc[r * 3 + c][0] += wv * q.x; acc[r * 3 + c][1] += wv * q.y;
                            acc[r * 3 + c][2] += wv * q.z; acc[r * 3 + c][3] += wv * q.w;
                        }
                    if (i + 3 < 18) {
#pragma unroll
                        for (int r = 0; r < 3; ++r) col[i % 3][r] = *reinterpret_cast<const float4*>(xs + rowbase + (r * p.TinW + i + 3) * XW);
                    }
                }
            } else {
                float win[3][3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 3; ++r) win[c][r] = xs[rowbase + (r * p.TinW + c) * XW + e];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float nv[NN];
#pragma unroll
                    for (int k = 0; k < NN; k += 4) {
                        const float4 q = *reinterpret_cast<const float4*>(dys + (slot8 * 16 + i) * DYW + k);
                        nv[k] = q.x; nv[k + 1] = q.y; nv[k + 2] = q.z; nv[k + 3] = q.w;
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float wv = win[(i + c) % 3][r];
#pragma unroll
                            for (int k = 0; k < NN; ++k) acc[r * 3 + c][k] += wv * nv[k];
                        }
                    if (i + 3 < 18) {
#pragma unroll
                        for (int r = 0; r < 3; ++r) win[i % 3][r] = xs[rowbase + (r * p.TinW + i + 3) * XW + e];
                    }
                }
            }
        } else
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int m = slot8 + 8 * i;
            const int ti = tab_in[m];
            if (NARROW_X) {
                const float wv = dys[m * DYW + e];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 q = *reinterpret_cast<const float4*>(xs + ti + toff[t]);
                    acc[t][0] += wv * q.x; acc[t][1] += wv * q.y; acc[t][2] += wv * q.z; acc[t][3] += wv * q.w;
                }
            } else {
                float nv[NN];
#pragma unroll
                for (int k = 0; k < NN; k += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(dys + m * DYW + k);
                    nv[k] = q.x; nv[k + 1] = q.y; nv[k + 2] = q.z; nv[k + 3] = q.w;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float wv = xs[ti + toff[t] + e];
#pragma unroll
                    for (int k = 0; k < NN; ++k) acc[t][k] += wv * nv[k];
                }
            }
        }
        __syncthreads();                                 // drains this wave's DMA (vmcnt(0)) and orders the buffers
        buf ^= 1;
    }
    float* red = xs0;                     // [8][NN][32] (>= 4 KB: the x double buffer)
    float* out = p.slab + (((long long)split * p.base + (bid - split * p.base)) * J) * 1024;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NN; ++k) red[(slot8 * NN + k) * 32 + e] = (wide_ok && ((tvalid >> t) & 1u)) ? acc[t][k] : 0.f;
        __syncthreads();
        if (t < J * p.TPS) {
            const int jj = t / p.TPS, tl = t - jj * p.TPS;
            for (int i = tid; i < NN * 32; i += 256) {
                const int k = i >> 5, ee = i & 31;
                float s_ = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) s_ += red[(q * NN + k) * 32 + ee];
                if (NARROW_X) { if (k < p.CW) out[jj * 1024 + (tl * p.CW + k) * 32 + ee] = s_; }
                else out[jj * 1024 + ee * 32 + k] = s_;
            }
        }
    }
    if (do_bias) {
        __syncthreads();
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 32 + tid];
            p.bias_slab[((long long)split * p.nCo + coc) * 32 + tid] = t;
        }
    }
}

// ---------------------------------------------------------------- 1x1 convolution with few channels on both sides
// The decoder head (16 -> 7, model.py:2568) has 112 weights and 2.1 M positions: its weight gradient is a streaming
// reduction of outer products (floor: reading x and dy once, ~40 us), for which the tiled kernels above spend their
// time on padding (157 us).  Tiles of 512 positions are staged with coalesced loads, thread (half, ci, co) walks half
// of the tile with two LDS reads (both mostly broadcasts) per FMA; sums stay in registers across the tiles of the
// workgroup and leave through the common slab layout ([32][32] block: row = ci, column = co).
#define WGP_TILE 512
__global__ __launch_bounds__(256) void wgrad_pointwise_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                              long long P, int Ci, int Co, float* __restrict__ slab,
                                                              float* __restrict__ bias_slab, int vec_x) {
    __shared__ float xs[WGP_TILE * 16];
    __shared__ float dys[WGP_TILE * 8];
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int half = tid >> 7, ci = (tid >> 3) & 15, co = tid & 7;
    const long long ntiles = (P + WGP_TILE - 1) / WGP_TILE;
    float acc = 0.f, bsum = 0.f;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long p0 = tile * WGP_TILE;
        __syncthreads();
        if (vec_x) {                                     // Ci == 16, 16-byte rows
            for (int i = tid; i < WGP_TILE * 4; i += 256) {
                const long long pp = p0 + (i >> 2);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pp < P) v = *reinterpret_cast<const float4*>(x + pp * ldx + 4 * (i & 3));
                *reinterpret_cast<float4*>(xs + 4 * i) = v;
            }
        } else {
            for (int i = tid; i < WGP_TILE * 16; i += 256) {
                const long long pp = p0 + (i >> 4);
                const int c = i & 15;
                xs[i] = (pp < P && c < Ci) ? x[pp * ldx + c] : 0.f;
            }
        }
        for (int i = tid; i < WGP_TILE * 8; i += 256) {
            const long long pp = p0 + (i >> 3);
            const int k = i & 7;
            dys[i] = (pp < P && k < Co) ? dy[pp * lddy + k] : 0.f;
        }
        __syncthreads();
        const float* xr = xs + half * (WGP_TILE / 2) * 16 + ci;
        const float* dr = dys + half * (WGP_TILE / 2) * 8 + co;
#pragma unroll 8
        for (int q = 0; q < WGP_TILE / 2; ++q) {
            const float d = dr[q * 8];
            acc += xr[q * 16] * d;
            if (ci == 0) bsum += d;
        }
    }
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    if (half == 0) slab[(long long)blockIdx.x * 1024 + ci * 32 + co] = red[tid] + red[tid + 128];
    __syncthreads();
    red[tid] = bsum;
    __syncthreads();
    if (bias_slab != nullptr && tid < 8) bias_slab[(long long)blockIdx.x * 32 + tid] = red[tid] + red[tid + 128];
}

// dw_tck[t][ci][co] = sum over the split-K slabs and dbias[co] = sum over the per-split column sums, ONE launch:
// block (64, SL): 64 consecutive outputs x SL slab lanes; lane y adds slabs y, y+SL, ... in order, the SL partial
// sums are then added in lane order -- a fixed summation tree, so the result is bit-reproducible (no atomics).
// Outputs [0, total) are weight elements, [total, total + Co) the bias gradient.
struct WgradTapMap { int slot[MRDIS_MAX_TAPS]; };    // tap row of dw_tck -> tap slot of the launch (stride-2: grouped by parity class)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ntaps, int Ci, int Co,
                                    int CW, int TPS, int J, int nCi, int nCo, int base, int nslab, WgradTapMap map,
                                    const float* __restrict__ bslab, float* __restrict__ dbias, int accumulate_bias) {
    __shared__ float red[16][65];
    const int total = ntaps * Ci * Co;                   // < 2^31 (host: kh*kw*Ci*Co elements of one filter)
    const int nout = total + (dbias != nullptr ? Co : 0);
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int SL = blockDim.y, y = threadIdx.y;
    float s_ = 0.f;
    if (i < total) {
        const int co = i % Co;
        const int r = i / Co;
        const int ci = r % Ci, t = r / Ci;
        const int coc = co >> 5, n = co & 31;
        const int cic = (Ci >= 32) ? (ci >> 5) : 0;
        const int cl = ci - cic * 32;
        const int sl = map.slot[t];
        const int jj = sl / TPS, m = (sl - jj * TPS) * CW + cl;
        const int g = jj / J, j = jj - g * J;
        const int b = (g * nCi + cic) * nCo + coc;
        const long long stride = (long long)base * J * 1024;
        const float* src = slab + ((long long)b * J + j) * 1024 + m * 32 + n + (long long)y * stride;
        const long long step = (long long)SL * stride;
        int k = y;
        for (; k + 3 * SL < nslab; k += 4 * SL, src += 4 * step) {        // four loads in flight, summed in slab order
            const float a = src[0], b = src[step], c = src[2 * step], d = src[3 * step];
            s_ += a; s_ += b; s_ += c; s_ += d;
        }
        for (; k < nslab; k += SL, src += step) s_ += *src;
    } else if (i < nout) {
        const int co = i - total;
        for (int k = y; k < nslab; k += SL) s_ += bslab[(long long)k * (nCo * 32) + co];
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

struct WgradPlan {
    WgradParams p;
    int J;
    size_t lds;
    long long slab_floats, part_floats, bias_floats;
    int nchunk;
    int dma, XR;                  // LDS-DMA kernel usable; staged x rows (multiple of 8)
    int out_taps;                 // kh*kw rows of dw_tck
    WgradTapMap map;
    int thin, thin_nt, thin_dma;  // 0 | 1 (narrow dy) | 2 (narrow x); tap slots per group; staging fits
    int pointwise;                // 1x1 s1 p0 with Ci <= 16, Co <= 8: streaming outer-product kernel

    size_t lds_dma;
};

// Tap groups.  A stride-1 layer has one input view and its kh*kw taps are cut into groups of J sub-tiles.
// A stride-2 layer is split into the four parity classes of the input (x[2a+p][2b+q]): inside a class every
// tap is a STRIDE-1 tap on the subsampled view, so the staged tile holds only pixels that are used (a
// stride-2 tile is 4x larger than what each tap reads), LDS reads are conflict-free and the LDS-DMA kernel
// applies (its double-buffered stride-2 tile would not fit).  One class = one tap group of the same launch.
static int plan_wgrad(WgradPlan& pl, int N, int H, int W, int ldx, int Ci, int Co, int kh, int kw, int stride, int pad) {
    int Ho, Wo;
    int rc = check_conv_geom(N, H, W, Ci, Co, kh, kw, stride, pad, &Ho, &Wo);
    if (rc) return rc;
    WgradParams& p = pl.p;
    p = WgradParams{};
    p.N = N; p.Hin = H; p.Win = W; p.Ci = Ci; p.A = Ho; p.B = Wo; p.Co = Co; p.is = 1;
    p.x_img = (long long)H * W * ldx;
    int CW = 32;
    if (Ci < 32) { CW = 4; while (CW < Ci) CW <<= 1; }
    p.CW = CW; p.TPS = 32 / CW;
    static const int JS[] = {1, 2, 3, 4, 5, 8, 9};
    auto round_J = [&](int J) { for (int k = 0; k < 7; ++k) if (JS[k] >= J) return JS[k]; return 9; };
    int J, ext_h = 0, ext_w = 0;                        // tile halo extents (max over groups)
    for (int t = 0; t < MRDIS_MAX_TAPS; ++t) pl.map.slot[t] = 0;
    if (stride == 1) {
        p.x_row = W * ldx; p.x_pix = ldx;
        const int subtiles = mrdis_cdiv(kh * kw, p.TPS);
        int nG = 1; J = subtiles;
        if (subtiles > 9) { nG = mrdis_cdiv(subtiles, 8); J = mrdis_cdiv(subtiles, nG); }
        J = round_J(J);
        p.nG = mrdis_cdiv(subtiles, J);
        if (p.nG > WG_MAX_GROUPS || p.nG * J * p.TPS > WG_MAX_SLOTS) return MRDIS_EUNSUPPORTED;
        for (int g = 0; g < p.nG; ++g) { p.g_off[g] = 0; p.g_Hin[g] = H; p.g_Win[g] = W; p.g_dhmin[g] = -pad; p.g_dwmin[g] = -pad; }
        for (int r = 0; r < kh; ++r) for (int s_ = 0; s_ < kw; ++s_) {
            const int t = r * kw + s_;
            p.dh[t] = r - pad; p.dw[t] = s_ - pad; p.slot_ok |= 1u << t; pl.map.slot[t] = t;
        }
        ext_h = kh - 1; ext_w = kw - 1;
        pl.thin_nt = (p.nG == 1) ? kh * kw : 99;
    } else {
        p.x_row = 2 * W * ldx; p.x_pix = 2 * ldx;
        int cnt[4] = {0, 0, 0, 0}, maxc = 0;
        for (int r = 0; r < kh; ++r) for (int s_ = 0; s_ < kw; ++s_) { const int c = (((r - pad) & 1) << 1) | ((s_ - pad) & 1); if (++cnt[c] > maxc) maxc = cnt[c]; }
        J = round_J(mrdis_cdiv(maxc, p.TPS));
        pl.thin_nt = maxc;
        if (mrdis_cdiv(maxc, p.TPS) > 9 || 4 * J * p.TPS > WG_MAX_SLOTS) return MRDIS_EUNSUPPORTED;
        p.nG = 4;
        int fill[4] = {0, 0, 0, 0};
        for (int c = 0; c < 4; ++c) {
            const int pp = c >> 1, qq = c & 1;
            p.g_off[c] = ((long long)pp * W + qq) * ldx;
            p.g_Hin[c] = (H - pp + 1) / 2; p.g_Win[c] = (W - qq + 1) / 2;
            p.g_dhmin[c] = 1 << 20; p.g_dwmin[c] = 1 << 20;
        }
        int dhmax[4] = {-(1 << 20), -(1 << 20), -(1 << 20), -(1 << 20)}, dwmax[4] = {-(1 << 20), -(1 << 20), -(1 << 20), -(1 << 20)};
        for (int r = 0; r < kh; ++r) for (int s_ = 0; s_ < kw; ++s_) {
            const int u = r - pad, v = s_ - pad, pp = u & 1, qq = v & 1, c = (pp << 1) | qq;
            const int slot = c * J * p.TPS + fill[c]++;
            p.dh[slot] = (u - pp) / 2; p.dw[slot] = (v - qq) / 2; p.slot_ok |= 1u << slot; pl.map.slot[r * kw + s_] = slot;
            if (p.dh[slot] < p.g_dhmin[c]) p.g_dhmin[c] = p.dh[slot];
            if (p.dw[slot] < p.g_dwmin[c]) p.g_dwmin[c] = p.dw[slot];
            if (p.dh[slot] > dhmax[c]) dhmax[c] = p.dh[slot];
            if (p.dw[slot] > dwmax[c]) dwmax[c] = p.dw[slot];
        }
        for (int c = 0; c < 4; ++c) {
            if (cnt[c] == 0) { p.g_dhmin[c] = 0; p.g_dwmin[c] = 0; continue; }
            if (dhmax[c] - p.g_dhmin[c] > ext_h) ext_h = dhmax[c] - p.g_dhmin[c];
            if (dwmax[c] - p.g_dwmin[c] > ext_w) ext_w = dwmax[c] - p.g_dwmin[c];
        }
    }
    pl.J = J;
    p.ntaps = p.nG * J * p.TPS;
    pl.out_taps = kh * kw;
    p.nCi = (Ci >= 32) ? mrdis_cdiv(Ci, 32) : 1;
    p.nCo = mrdis_cdiv(Co, 32);
    p.base = p.nCi * p.nCo * p.nG;
    // position tile: shrink until LDS fits (<= 64 KiB -> two workgroups per CU)
    TileChoice tc = choose_tile(N, Ho, Wo);
    for (;;) {
        p.NB = tc.NB; p.TH = tc.TH; p.TW = tc.TW;
        p.TinH = (p.TH - 1) * p.is + ext_h + 1;
        p.TinW = (p.TW - 1) * p.is + ext_w + 1;
        pl.lds = sizeof(float) * (256 + 128 * 32 + (size_t)p.NB * p.TinH * p.TinW * (CW + 1));
        if (pl.lds <= 64 * 1024) break;
        if (tc.TH > 1) tc.TH = (tc.TH + 1) / 2; else if (tc.NB > 1) tc.NB = (tc.NB + 1) / 2; else if (tc.TW > 1) tc.TW = (tc.TW + 1) / 2; else return MRDIS_EUNSUPPORTED;
    }
    // LDS-DMA variant: full 32-channel chunks on both sides, J taps per group all real, and a double-buffered
    // tile pair that still lets two workgroups share a CU (<= 80 KB each)
    pl.dma = 0; pl.XR = 0; pl.lds_dma = 0;
    if (Ci % 32 == 0 && Co % 4 == 0 && p.TPS == 1 && !mrdis_opt(MRDIS_OPT_NODMA)) {
        TileChoice cand[2] = {tc, TileChoice{1, 8, 16}};
        for (int c = 0; c < 2 && !pl.dma; ++c) {
            const TileChoice t = cand[c];
            if (c == 1 && !(Ho % 8 == 0 && Wo % 16 == 0)) break;
            const int tinH = (t.TH - 1) * p.is + ext_h + 1, tinW = (t.TW - 1) * p.is + ext_w + 1;
            const int npix = t.NB * tinH * tinW, XR = (npix + 7) / 8 * 8;
            const size_t lds = sizeof(float) * (128 + 2 * 128 * 32 + 2 * (size_t)XR * 32);
            if (lds <= 80 * 1024 && XR / 8 <= 4 * WGD_XSLOTS && t.NB < 1024 && tinH < 1024 && tinW < 1024) {
                pl.dma = 1; pl.XR = XR; pl.lds_dma = lds;
                p.NB = t.NB; p.TH = t.TH; p.TW = t.TW; p.TinH = tinH; p.TinW = tinW;
                pl.lds = sizeof(float) * (256 + 128 * 32 + (size_t)npix * (CW + 1));   // same tile if the plain kernel has to run
                if (pl.lds > 64 * 1024) { pl.dma = 0; }
            }
        }
    }
    // thin product (one side 4 channels wide): packed FMAs instead of a padded MFMA tile.  Worth it only on the
    // large maps (launch + reduce overheads dominate the small ones either way).
    {
        const int want = pl.thin_nt;
        pl.thin = 0; pl.thin_nt = 0; pl.thin_dma = 0;
        const bool big = (long long)N * Ho * Wo >= 100000;
        if (big && !mrdis_opt(MRDIS_OPT_NOTHIN) && !mrdis_opt(MRDIS_OPT_NODMA) && Co % 4 == 0 && p.NB < 1024 && p.TinH < 1024 && p.TinW < 1024) {
            if (Co == 4 && Ci % 32 == 0 && want == 9) { pl.thin = 1; pl.thin_nt = 9; }
            else if (Ci == 4 && p.CW == 4 && Co >= 16 && want <= 9) { pl.thin = 2; pl.thin_nt = want <= 4 ? 4 : 9; }
        }
        if (pl.thin) {
            const int npix = p.NB * p.TinH * p.TinW;
            const int rpi = pl.thin == 2 ? 64 : 8, xw = pl.thin == 2 ? 4 : 32, dyw = pl.thin == 2 ? 32 : 4;
            const int XR = (npix + rpi - 1) / rpi * rpi;
            size_t lds = sizeof(float) * (128 + 2 * 128 * dyw + 2 * (size_t)XR * xw);
            const size_t lds_min = sizeof(float) * (128 + 2 * 128 * dyw + 8 * 4 * 32 + 256);    // epilogue scratch
            if (lds < lds_min) lds = lds_min;
            if (lds <= 80 * 1024 && XR / rpi <= 4 * WGD_XSLOTS) { pl.thin_dma = 1; pl.XR = XR; pl.lds_dma = lds; pl.dma = 0; }
            else pl.thin = 0;
        }
    }
    p.tilesA = mrdis_cdiv(Ho, p.TH); p.tilesB = mrdis_cdiv(Wo, p.TW); p.tilesN = mrdis_cdiv(N, p.NB);
    const long long nt = (long long)p.tilesA * p.tilesB * p.tilesN;
    if (nt > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.numTiles = (int)nt;
    // split-K: one workgroup per resident slot.  The MFMA kernels hold two workgroups per CU (accumulators), so 512
    // workgroups run as one full round and write half the slab volume a 1024-workgroup launch would
    // (1-7 % faster on the big layers, tools/ab_lib.py); the thin FMA kernels fit 3-4 per CU and keep 1024
    int wg_target = pl.thin ? 1024 : 512;
    if (mrdis_opt(MRDIS_OPT_WGSPLIT) >= 0) wg_target = (int)mrdis_opt(MRDIS_OPT_WGSPLIT);
    long long splits = mrdis_cdiv(wg_target, p.base);
    if (splits > nt) splits = nt;
    if (splits < 1) splits = 1;
    p.splits = (int)splits;
    pl.pointwise = 0;
    if (kh == 1 && kw == 1 && stride == 1 && pad == 0 && Ci <= 16 && Co <= 8 && p.CW == 16 && p.nG == 1 && J == 1 &&
        (long long)N * Ho * Wo >= 50000 && !mrdis_opt(MRDIS_OPT_NOTHIN)) {
        pl.pointwise = 1; pl.dma = 0; pl.thin = 0;
        long long tiles = ((long long)N * Ho * Wo + WGP_TILE - 1) / WGP_TILE;
        p.splits = (int)(tiles < 768 ? tiles : 768);
    }
    pl.slab_floats = (long long)p.splits * p.base * J * 1024;
    pl.nchunk = 1;
    pl.part_floats = 0;
    pl.bias_floats = (long long)p.splits * p.nCo * 32;
    return MRDIS_OK;
}

static size_t wgrad_need(const WgradPlan& pl) {
    return sizeof(float) * (size_t)(pl.slab_floats + pl.part_floats + pl.bias_floats) + 256;
}

size_t mrdis_wgrad16_workspace(int N, int H, int W, int Ci, int Co);            // mrdis_wgrad16.hip: Cout <= 16, 3x3 s1 p1
size_t mrdis_wgrad_s2_workspace(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad);     // mrdis_wgrad_s2.hip: Cin <= 7 stride-2 first layers
size_t mrdis_wgrad_c4_workspace(int N, int H, int W, int Ci, int Co);              // mrdis_wgrad_s2.hip: the 4 -> C si_layers
size_t mrdis_wgrad_co4b_workspace(int N, int H, int W, int Ci, int Co);
int mrdis_run_wgrad_co4b(const void* x_bf16, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                         int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s, int pad16);
int mrdis_run_wgrad_c4(const float* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                       int N, int H, int W, int Ci, int Co, int accumulate_bias, int dy_bf16, hipStream_t s, int pad16 = 0);
size_t mrdis_pw_wgrad_workspace(long long npix, int Ci, int Co, int x16_bf16);      // mrdis_pointwise.hip: the 1x1 16 -> <= 8 head
int mrdis_run_pw_wgrad(const void* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                       long long npix, int Ci, int Co, int accumulate_bias, int x16_bf16, hipStream_t s);
int mrdis_run_wgrad_s2(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                       int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int accumulate_bias, hipStream_t s);
int mrdis_run_wgrad16_bf16(const void* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s);      // mrdis_wgrad16.hip: bf16 views, 32 -> 16
int mrdis_run_wgrad16(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                      size_t workspace_bytes, int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s);
size_t mrdis_wino_wgrad_workspace(int N, int H, int W, int Ci, int Co);
int mrdis_run_wino_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                         size_t workspace_bytes, int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s);
// Winograd weight gradient (mrdis_wino.hip): 3x3 s1 p1 layers with Ci, Co multiples of 32 / 64 and enough tiles per split
static bool wino_wgrad_wanted(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad) {
    const int mode = (int)mrdis_opt(MRDIS_OPT_WINO);          // MRDIS_WINO at load; mrdis_set_option("wino", v) afterwards
    if (mode == 0 || kh != 3 || kw != 3 || stride != 1 || pad != 1) return false;
    if (mrdis_wino_wgrad_workspace(N, H, W, Ci, Co) == 0) return false;
    if (mode == 2) return true;
    // measured (tools/layer_bench.py, B = 32): 1.3-1.7x on every 64 x 64 and 32 x 64 blocked layer down to 16x16 maps; the
    // 64 x 32 blocking (Cout = 32) amortises the input transform over too few couts unless Cin >= 64 (sp5.out 64 -> 32: 195 -> 138 us)
    if (Co % 64 != 0 && Ci < 64) return false;
    return (long long)N * ((H + 3) / 4) * ((W + 7) / 8) >= 256;
}

// mrdis_bf16.hip: weight gradient on bf16 MFMA operands (stride-1 "same" convolutions, Ci % 32 == 0)
size_t mrdis_bwgrad_workspace(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad);
int mrdis_run_bwgrad(const void* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                     int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int accumulate_bias, int dtype, hipStream_t s);

extern "C" size_t mrdis_conv2d_bwd_weight_workspace(int N, int H, int W, int Ci, int Co,
                                                    int kh, int kw, int stride, int pad) {
    WgradPlan pl;
    if (plan_wgrad(pl, N, H, W, Ci, Ci, Co, kh, kw, stride, pad)) return 0;
    size_t need = wgrad_need(pl);
    { const size_t nb = mrdis_bwgrad_workspace(N, H, W, Ci, Co, kh, kw, stride, pad); if (nb > need) need = nb; }   // any dtype may be asked for
    if (kh == 3 && kw == 3 && stride == 1 && pad == 1) {           // either kernel may run (MRDIS_WINO is read per call)
        const size_t nw = mrdis_wino_wgrad_workspace(N, H, W, Ci, Co);
        if (nw > need) need = nw;
        const size_t n16 = mrdis_wgrad16_workspace(N, H, W, Ci, Co);
        if (n16 > need) need = n16;
        const size_t n4 = mrdis_wgrad_c4_workspace(N, H, W, Ci, Co);
        if (n4 > need) need = n4;
        const size_t n4b = mrdis_wgrad_co4b_workspace(N, H, W, Ci, Co);
        if (n4b > need) need = n4b;
    }
    { const size_t n2 = mrdis_wgrad_s2_workspace(N, H, W, Ci, Co, kh, kw, stride, pad); if (n2 > need) need = n2; }
    if (kh == 1 && kw == 1 && stride == 1 && pad == 0) { const size_t n1 = mrdis_pw_wgrad_workspace((long long)N * H * W, Ci, Co, 1); if (n1 > need) need = n1; }
    return need;
}

template <int J>
static int launch_wgrad_t(const WgradPlan& pl, hipStream_t s) {
    if (pl.dma) {
        static bool attr_set = false;
        if (!attr_set) {      // > 64 KB of dynamic LDS needs the opt-in once per instantiation
            if (hipFuncSetAttribute((const void*)wgrad_dma_kernel<J>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
                return MRDIS_ELAUNCH;
            attr_set = true;
        }
        const bool x_ok = (pl.p.ldx % 4 == 0) && (((uintptr_t)pl.p.x & 15) == 0);
        const bool dy_ok = (pl.p.lddy % 4 == 0) && (((uintptr_t)pl.p.dy & 15) == 0);
        if (x_ok && dy_ok) {
            MRDIS_LAUNCH((wgrad_dma_kernel<J>), dim3(pl.p.splits * pl.p.base), dim3(256), pl.lds_dma, s, pl.p, pl.XR);
            MRDIS_CHECK_LAUNCH();
            return MRDIS_OK;
        }
    }
    MRDIS_LAUNCH((wgrad_kernel<J>), dim3(pl.p.splits * pl.p.base), dim3(256), pl.lds, s, pl.p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_conv2d_bwd_weight(const void* x_, int ldx, const void* dy_, int lddy,
                                       float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                                       int N, int H, int W, int Ci, int Co,
                                       int kh, int kw, int stride, int pad, int accumulate_bias, int dtype, void* stream) {
    // MRDIS_DT_DW_PAD16 (mixed-storage calls only): dw_tck has the stored shape of a filter of the mixing launch -- [T][16][Co] for the 4 -> C layers,
    // [T][Ci][16] for the C -> 4 layer -- and the rows / columns beyond the layer's own are written as zeros
    const int pad16 = (dtype & MRDIS_DT_DW_PAD16) ? 1 : 0;
    dtype &= ~MRDIS_DT_DW_PAD16;
    if (pad16 && dtype != MRDIS_DT_XBF16_YF32 && dtype != MRDIS_DT_XF32_YBF16) return MRDIS_EUNSUPPORTED;
    if (dtype != MRDIS_DT_F32 && dtype != MRDIS_DT_F32_BF16M && dtype != MRDIS_DT_BF16 && dtype != MRDIS_DT_XBF16_YF32 && dtype != MRDIS_DT_XF32_YBF16) return MRDIS_EUNSUPPORTED;
    const float* x = reinterpret_cast<const float*>(x_); const float* dy = reinterpret_cast<const float*>(dy_);
    WgradPlan pl;
    int rc = plan_wgrad(pl, N, H, W, ldx, Ci, Co, kh, kw, stride, pad);
    if (rc) return rc;
    if (!x || !dy || !dw_tck || !workspace || ldx < Ci || lddy < Co) return MRDIS_EINVAL;
    if (workspace_bytes < wgrad_need(pl)) return MRDIS_EWORKSPACE;
    if (((uintptr_t)workspace & 15) != 0) return MRDIS_EALIGN;
    if (dtype == MRDIS_DT_XBF16_YF32) {       // the 1x1 head under bf16 storage: x bf16 (16 channels), dy fp32 (<= 8 channels)
        if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && Co == 4)      // ana_dec.output: C -> 4, the bf16 trunk x the fp32 gradient of the anatomy logits
            return mrdis_run_wgrad_co4b(x_, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, accumulate_bias, (hipStream_t)stream, pad16);
        if (!(kh == 1 && kw == 1 && stride == 1 && pad == 0) || pad16) return MRDIS_EUNSUPPORTED;
        return mrdis_run_pw_wgrad(x_, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, (long long)N * H * W, Ci, Co, accumulate_bias, 1, (hipStream_t)stream);
    }
    if (dtype == MRDIS_DT_XF32_YBF16) {       // the 4 -> C si_layers under bf16 storage: x the fp32 anatomy map, dy bf16
        if (!(kh == 3 && kw == 3 && stride == 1 && pad == 1 && Ci == 4)) return MRDIS_EUNSUPPORTED;
        return mrdis_run_wgrad_c4(x, ldx, dy_, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, accumulate_bias, 1, (hipStream_t)stream, pad16);
    }
    if (dtype == MRDIS_DT_BF16 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Ci == 32 && Co == 16) {      // sp6.out on bf16 views (mrdis_wgrad16.hip)
        rc = mrdis_run_wgrad16_bf16(x_, ldx, dy_, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, accumulate_bias, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (dtype == MRDIS_DT_F32_BF16M || dtype == MRDIS_DT_BF16) {
        rc = mrdis_run_bwgrad(x_, ldx, dy_, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, kh, kw, stride, pad,
                              accumulate_bias, dtype, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED || dtype == MRDIS_DT_BF16) return rc;          // bf16 views never reach the fp32 kernels
    }
    if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && Ci == 4) {
        rc = mrdis_run_wgrad_c4(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, accumulate_bias, 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (kh == 3 && kw == 3 && stride == 1 && pad == 1 && Co <= 16) {
        rc = mrdis_run_wgrad16(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, accumulate_bias, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (kh == 1 && kw == 1 && stride == 1 && pad == 0) {
        rc = mrdis_run_pw_wgrad(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, (long long)N * H * W, Ci, Co, accumulate_bias, 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (stride == 2 && Ci <= 7) {
        rc = mrdis_run_wgrad_s2(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, accumulate_bias,
                                (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (wino_wgrad_wanted(N, H, W, Ci, Co, kh, kw, stride, pad)) {
        rc = mrdis_run_wino_wgrad(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, accumulate_bias, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    WgradParams& p = pl.p;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy;
    p.slab = reinterpret_cast<float*>(workspace);
    float* part = p.slab + pl.slab_floats;
    p.bias_slab = dbias ? part + pl.part_floats : nullptr;
    p.vec_x = (Ci % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0) && (p.CW % 4 == 0);
    p.vec_dy = (Co % 4 == 0) && (lddy % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    const bool x_al = (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0), dy_al = (lddy % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    if (pl.thin && pl.thin_dma && x_al && dy_al) {
        static bool attr_set = false;
        if (!attr_set) {      // > 64 KB of dynamic LDS needs the opt-in
            if (hipFuncSetAttribute((const void*)wgrad_thin_dma_kernel<9, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
                return MRDIS_ELAUNCH;
            attr_set = true;
        }
        if (pl.thin == 1) MRDIS_LAUNCH((wgrad_thin_dma_kernel<9, 4, false>), dim3(p.splits * p.base), dim3(256), pl.lds_dma, s, p, pl.J, pl.XR);
        else if (pl.thin_nt == 4) MRDIS_LAUNCH((wgrad_thin_dma_kernel<4, 4, true>), dim3(p.splits * p.base), dim3(256), pl.lds_dma, s, p, pl.J, pl.XR);
        else MRDIS_LAUNCH((wgrad_thin_dma_kernel<9, 4, true>), dim3(p.splits * p.base), dim3(256), pl.lds_dma, s, p, pl.J, pl.XR);
        MRDIS_CHECK_LAUNCH();
        rc = MRDIS_OK;
    } else
    if (pl.pointwise) {
        const int vx = (Ci == 16) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
        MRDIS_LAUNCH(wgrad_pointwise_kernel, dim3(p.splits), dim3(256), 0, s, x, ldx, dy, lddy, (long long)N * H * W, Ci, Co,
                           p.slab, p.bias_slab, vx);
        MRDIS_CHECK_LAUNCH();
        rc = MRDIS_OK;
    } else
    switch (pl.J) {
        case 1: rc = launch_wgrad_t<1>(pl, s); break;
        case 2: rc = launch_wgrad_t<2>(pl, s); break;
        case 3: rc = launch_wgrad_t<3>(pl, s); break;
        case 4: rc = launch_wgrad_t<4>(pl, s); break;
        case 5: rc = launch_wgrad_t<5>(pl, s); break;
        case 8: rc = launch_wgrad_t<8>(pl, s); break;
        case 9: rc = launch_wgrad_t<9>(pl, s); break;
        default: return MRDIS_EUNSUPPORTED;
    }
    if (rc) return rc;
    const long long nout = (long long)pl.out_taps * Ci * Co + (dbias ? Co : 0);
    int SL = 1;
    while (SL < 16 && SL * 8 <= p.splits) SL <<= 1;           // >= 8 slabs per lane before another lane is added
    MRDIS_LAUNCH(wgrad_reduce_kernel, dim3(mrdis_cdiv(nout, 64)), dim3(64, SL), 0, s, p.slab, dw_tck, pl.out_taps, Ci, Co,
                       p.CW, p.TPS, pl.J, p.nCi, p.nCo, p.base, p.splits, pl.map, p.bias_slab, dbias, accumulate_bias ? 1 : 0);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// =========================================================================== expert mixing
__global__ void mix_fwd_kernel(const float* __restrict__ W, const float* __restrict__ r, float* __restrict__ w_tck,
                               float* __restrict__ w_tkc, int E, int Co, int Ci, int T) {
    const long long total = (long long)Co * Ci * T;
    float rr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) rr[e] = e < E ? r[e] : 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const long long q = i / T;
        const int ci = (int)(q % Ci), co = (int)(q / Ci);
        float s = 0.f;
        for (int e = 0; e < E; ++e) s += rr[e] * W[(long long)e * total + i];   // sum over experts in order e = 0..E-1
        if (w_tck) w_tck[((long long)t * Ci + ci) * Co + co] = s;
        if (w_tkc) w_tkc[((long long)t * Co + co) * Ci + ci] = s;
    }
}

__global__ void mix_bwd_kernel(const float* __restrict__ dw_tck, const float* __restrict__ W, const float* __restrict__ r,
                               float* __restrict__ dW, float* __restrict__ part, int E, int Co, int Ci, int T) {
    __shared__ double red[8][4];
    const long long total = (long long)Co * Ci * T;
    float rr[8];
    double dr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { rr[e] = e < E ? r[e] : 0.f; dr[e] = 0.0; }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % T);
        const long long q = i / T;
        const int ci = (int)(q % Ci), co = (int)(q / Ci);
        const float g = dw_tck[((long long)t * Ci + ci) * Co + co];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < E) {
                dW[(long long)e * total + i] = rr[e] * g;
                dr[e] += (double)g * (double)W[(long long)e * total + i];
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double v = mrdis_wave_sum_d(dr[e]);
        if (lane == 0) red[e][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8 && (int)threadIdx.x < E)
        part[(long long)blockIdx.x * 8 + threadIdx.x] = (float)((red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}
__global__ void mix_bwd_final_kernel(const float* __restrict__ part, int nblk, int E, float* __restrict__ dr) {
    const int e = threadIdx.x;
    if (e >= E) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(long long)b * 8 + e];
    dr[e] += (float)s;
}

// ---- routed variants: r = sigmoid(fc_w @ type + fc_b) (model.py:2071-2073) is evaluated inside the mix kernel
// and its backward inside the mix-backward finalize, so a CondConv2d call costs two launches instead of
// two plus ~9 tiny torch kernels (addmm, sigmoid, their backwards, zero fills).
__global__ void mix_routed_fwd_kernel(const float* __restrict__ W, const float* __restrict__ fcw, const float* __restrict__ fcb,
                                      const float* __restrict__ t, int emb, float* __restrict__ r_out,
                                      float* __restrict__ w_tck, float* __restrict__ w_tkc, int E, int Co, int Ci, int T) {
    const long long total = (long long)Co * Ci * T;
    float rr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float z = 0.f;
        if (e < E) { z = fcb[e]; for (int k = 0; k < emb; ++k) z += fcw[e * emb + k] * t[k]; }
        rr[e] = e < E ? 1.f / (1.f + expf(-z)) : 0.f;
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < E) r_out[threadIdx.x] = rr[threadIdx.x];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int tt = (int)(i % T);
        const long long q = i / T;
        const int ci = (int)(q % Ci), co = (int)(q / Ci);
        float s_ = 0.f;
        for (int e = 0; e < E; ++e) s_ += rr[e] * W[(long long)e * total + i];
        w_tck[((long long)tt * Ci + ci) * Co + co] = s_;
        w_tkc[((long long)tt * Co + co) * Ci + ci] = s_;
    }
}
// dfcw[e][k] = dr[e] r[e](1-r[e]) t[k], dfcb[e] = dr[e] r[e](1-r[e])
__global__ void mix_routed_bwd_final_kernel(const float* __restrict__ part, int nblk, int E, const float* __restrict__ r,
                                            const float* __restrict__ t, int emb, float* __restrict__ dfcw, float* __restrict__ dfcb) {
    const int e = threadIdx.x;
    if (e >= E) return;
    double s_ = 0.0;
    for (int b = 0; b < nblk; ++b) s_ += (double)part[(long long)b * 8 + e];
    const float dz = (float)s_ * r[e] * (1.f - r[e]);
    dfcb[e] = dz;
    for (int k = 0; k < emb; ++k) dfcw[e * emb + k] = dz * t[k];
}

// ---- all modality types of a layer at once.  A step uses every CondConv2d with each of the M type rows
// (model.py:3138: inputs_type = (1+i) * ones); mixing them in one forward and one backward launch pair cuts the
// launches 4x and removes the gradient accumulation adds autograd would issue for W / fc.weight / fc.bias.
#define MIX_MAX_TYPES 8
struct MixPtrs { float* tck[MIX_MAX_TYPES]; float* tkc[MIX_MAX_TYPES]; __bf16* btck[MIX_MAX_TYPES]; __bf16* btkc[MIX_MAX_TYPES]; };   // b*: optional bf16 copies
struct MixCPtrs { const float* p[MIX_MAX_TYPES]; };
// (bx, gx): block index and block count of this filter's share of the grid -- blockIdx.x / gridDim.x for the one-layer launch, the
// job's block range for the all-layers launch (mix_jobs_fwd_kernel)
// ---- tiled backward (round 6).  W is [E][Co][Ci][T] (the reference's parameter layout), the gradients of the mixed filters arrive as [T][Ci][Co]: walked
// element by element in W order, every gradient read is a 4-byte gather (16 x the sectors), and every type's row of blocks reads W again.  In the tiled form
// a thread owns a (cout, channel) pair, i.e. T contiguous parameters per expert (a wave touches 8 rows of 8 T floats), and the gradients come through an LDS
// tile filled in 128-byte rows.  Same sums in the same order as the element-wise body below (which remains for E > 3, M > 4 and other tap counts).  The
// forward mix was tried in the same form (LDS tile, 128-byte rows on both output layouts) and was no faster than its scattered stores (75 vs 66 us per
// launch): L2 merges those; it stays element-wise.
constexpr int MIXT_E = 3, MIXT_M = 4;      // (the model: 3 experts, <= 4 modality types; anything larger takes the element-wise bodies)
__device__ __forceinline__ bool mix_tiled_ok(int E, int M, int T) { return E <= MIXT_E && M <= MIXT_M && (T == 1 || T == 9 || T == 16); }

__device__ __forceinline__ void mix_fwd_body(const float* __restrict__ W, const float* __restrict__ fcw, const float* __restrict__ fcb,
                                             const float* __restrict__ types, int emb, float* __restrict__ r_out,
                                             float* __restrict__ w_tck, float* __restrict__ w_tkc, __bf16* __restrict__ b_tck, __bf16* __restrict__ b_tkc,
                                             int E, int Co, int Ci, int T, int ld_tck, long long tap_tkc, int m, int bx, int gx, int cip) {
    // cip: channel pitch of the outputs (Ci, or more when the filter is written into a zero-padded [T][cip][.] / [T][.][cip] buffer)
    // ld_tck: row pitch of the [T][Ci][.] outputs, tap_tkc: tap pitch of the [T][.][Ci] outputs -- Co and Co * Ci for a filter of its
    // own, 2 Co and 2 Co * Ci when the pointers address one half of a fused gamma | beta filter
    const float* t = types + m * emb;
    const long long total = (long long)Co * Ci * T;
    float rr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float z = 0.f;
        if (e < E) { z = fcb[e]; for (int k = 0; k < emb; ++k) z += fcw[e * emb + k] * t[k]; }
        rr[e] = e < E ? 1.f / (1.f + expf(-z)) : 0.f;
    }
    if (bx == 0 && (int)threadIdx.x < E) r_out[m * E + threadIdx.x] = rr[threadIdx.x];
    for (long long i = bx * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gx * blockDim.x) {
        const int tt = (int)(i % T);
        const long long q = i / T;
        const int ci = (int)(q % Ci), co = (int)(q / Ci);
        float s_ = 0.f;
        for (int e = 0; e < E; ++e) s_ += rr[e] * W[(long long)e * total + i];      // same order as the single-type kernel
        const long long o_tck = ((long long)tt * cip + ci) * ld_tck + co, o_tkc = (long long)tt * tap_tkc + (long long)co * cip + ci;
        w_tck[o_tck] = s_;
        w_tkc[o_tkc] = s_;
        if (b_tck != nullptr) {                              // the bf16 MFMA operands of compute_dtype bf16 (round to nearest even, as mrdis_cast_bf16)
            b_tck[o_tck] = (__bf16)s_;
            b_tkc[o_tkc] = (__bf16)s_;
        }
    }
}
__global__ void mix_routed_multi_fwd_kernel(const float* __restrict__ W, const float* __restrict__ fcw, const float* __restrict__ fcb,
                                            const float* __restrict__ types, int emb, float* __restrict__ r_out, MixPtrs out,
                                            int E, int Co, int Ci, int T, int ld_tck, long long tap_tkc) {
    const int m = blockIdx.y;
    mix_fwd_body(W, fcw, fcb, types, emb, r_out, out.tck[m], out.tkc[m], out.btck[m], out.btkc[m], E, Co, Ci, T, ld_tck, tap_tkc, m,
                 (int)blockIdx.x, (int)gridDim.x, Ci);
}
// block (b, m): partial dr[m][e] = <dw_m, W[e]> over the block's elements; the m == 0 blocks also write
// dW[e] = sum_m r[m][e] dw_m (types in order m = 0..M-1; a type without gradient contributes nothing).
// one block per tile does ALL types: W is read once (not once per type), every gradient once (not twice), in 128-byte rows through a double-buffered LDS
// tile (one barrier per gradient); dW[e] = sum_m r[m][e] dw_m in the same order as below: bit-identical; the routing dots in double as below (other order of
// the partial sums: equal to ~1e-16 relative before they are rounded to fp32)
template <int T>
__device__ __forceinline__ void mix_bwd_tiled(const MixCPtrs& dw, const float* __restrict__ W, const float* __restrict__ r, float* __restrict__ dW,
                                              float* __restrict__ part, int M, int E, int Co, int Ci, int accumulate, int ld_dw, int bx, int gx, int cip,
                                              double (*red)[4], float* __restrict__ gt_) {      // gt_: 2 x 9 x 8 x 33 floats of LDS, [buffer][tap][channel 8][cout 32 (+1)]
    constexpr int TP = T > 9 ? 8 : T;                 // taps per pass (16 taps: two passes of 8 -- the per-thread W chunk and sums stay in registers)
    const long long total = (long long)Co * Ci * T;
    const int tid = threadIdx.x, tco = tid >> 3, tci = tid & 7, sco = tid & 31, sci = tid >> 5;
    const int tilesI = (Ci + 7) / 8, ntiles = ((Co + 31) / 32) * tilesI;
    double dr[MIXT_M][MIXT_E];
    float rr[MIXT_M][MIXT_E];
#pragma unroll
    for (int mm = 0; mm < MIXT_M; ++mm)
#pragma unroll
        for (int e = 0; e < MIXT_E; ++e) { dr[mm][e] = 0.0; rr[mm][e] = (mm < M && e < E) ? r[mm * E + e] : 0.f; }
    int buf = 0;
    for (int tl = bx; tl < ntiles; tl += gx) {
        const int co0 = (tl / tilesI) * 32, ci0 = (tl % tilesI) * 8;
        const int co = co0 + tco, ci = ci0 + tci;
        const bool own = co < Co && ci < Ci, sok = co0 + sco < Co && ci0 + sci < Ci;
        const long long i0 = ((long long)co * Ci + ci) * T;
#pragma unroll
        for (int t0 = 0; t0 < T; t0 += TP) {
            float wv[MIXT_E][TP], acc[MIXT_E][TP];
#pragma unroll
            for (int e = 0; e < MIXT_E; ++e)
#pragma unroll
                for (int t = 0; t < TP; ++t) { wv[e][t] = (own && e < E) ? W[(long long)e * total + i0 + t0 + t] : 0.f; acc[e][t] = 0.f; }
#pragma unroll
            for (int mm = 0; mm < MIXT_M; ++mm) {
                if (mm >= M) break;
                const float* __restrict__ gp = dw.p[mm];
                if (gp == nullptr) continue;           // (block-uniform) a type without gradient contributes nothing
                float* L = gt_ + buf * (9 * 8 * 33);
#pragma unroll
                for (int t = 0; t < TP; ++t) L[(t * 8 + sci) * 33 + sco] = sok ? gp[((long long)(t0 + t) * cip + ci0 + sci) * ld_dw + co0 + sco] : 0.f;
                __syncthreads();
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    const float g = L[(t * 8 + tci) * 33 + tco];
#pragma unroll
                    for (int e = 0; e < MIXT_E; ++e) {
                        dr[mm][e] += (double)g * (double)wv[e][t];
                        acc[e][t] += rr[mm][e] * g;
                    }
                }
                buf ^= 1;                              // the next gradient lands in the other buffer: nobody still reads it (everyone passed the barrier above after its reads of two gradients ago)
            }
            if (own)
#pragma unroll
                for (int e = 0; e < MIXT_E; ++e)
                    if (e < E)
#pragma unroll
                        for (int t = 0; t < TP; ++t) { float* d_ = dW + (long long)e * total + i0 + t0 + t; *d_ = accumulate ? *d_ + acc[e][t] : acc[e][t]; }      // accumulate: dW is a gradient sink
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int mm = 0; mm < MIXT_M; ++mm) {
        if (mm >= M) break;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MIXT_E; ++e) {
            const double v = mrdis_wave_sum_d(dr[mm][e]);
            if (lane == 0) red[e][wave] = v;
        }
        __syncthreads();
        if (tid < 8 && tid < E) part[((long long)mm * gx + bx) * 8 + tid] = (float)((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]));
    }
}

__device__ __forceinline__ void mix_bwd_body(const MixCPtrs& dw, const float* __restrict__ W, const float* __restrict__ r,
                                             float* __restrict__ dW, float* __restrict__ part, int M, int E, int Co, int Ci, int T, int accumulate, int ld_dw,
                                             int m, int bx, int gx, int cip) {
    __shared__ double red[8][4];
    if (mix_tiled_ok(E, M, T) && blockDim.x == 256) {
        if (m != 0) return;                            // (the grid still has one row of blocks per type)
        __shared__ float gt[2 * 9 * 8 * 33];
        if (T == 9) mix_bwd_tiled<9>(dw, W, r, dW, part, M, E, Co, Ci, accumulate, ld_dw, bx, gx, cip, red, gt);
        else if (T == 16) mix_bwd_tiled<16>(dw, W, r, dW, part, M, E, Co, Ci, accumulate, ld_dw, bx, gx, cip, red, gt);
        else mix_bwd_tiled<1>(dw, W, r, dW, part, M, E, Co, Ci, accumulate, ld_dw, bx, gx, cip, red, gt);
        return;
    }
    const long long total = (long long)Co * Ci * T;
    const float* __restrict__ g_m = dw.p[m];
    double dr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dr[e] = 0.0;
    if (g_m != nullptr || m == 0)
    for (long long i = bx * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gx * blockDim.x) {
        const int t = (int)(i % T);
        const long long q = i / T;
        const int ci = (int)(q % Ci), co = (int)(q / Ci);
        const long long idx = ((long long)t * cip + ci) * ld_dw + co;         // ld_dw: row pitch of the gradient tensors (Co, or 2 Co for a fused half); cip: their channel pitch
        if (g_m != nullptr) {
            const float g = g_m[idx];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < E) dr[e] += (double)g * (double)W[(long long)e * total + i];
        }
        if (m == 0) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int mm = 0; mm < M; ++mm) {
                const float* gp = dw.p[mm];
                if (gp == nullptr) continue;
                const float g = gp[idx];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e < E) acc[e] += r[mm * E + e] * g;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < E) { float* d_ = dW + (long long)e * total + i; *d_ = accumulate ? *d_ + acc[e] : acc[e]; }      // accumulate: dW is a gradient sink
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const double v = mrdis_wave_sum_d(dr[e]);
        if (lane == 0) red[e][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8 && (int)threadIdx.x < E)
        part[((long long)m * gx + bx) * 8 + threadIdx.x] =
            (float)((red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}
__global__ __launch_bounds__(256) void mix_multi_bwd_kernel(MixCPtrs dw, const float* __restrict__ W, const float* __restrict__ r,
                                     float* __restrict__ dW, float* __restrict__ part, int M, int E, int Co, int Ci, int T, int accumulate, int ld_dw) {
    mix_bwd_body(dw, W, r, dW, part, M, E, Co, Ci, T, accumulate, ld_dw, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, Ci);
}
// one block of 64 threads: thread (m, e) sums its partials, dz = dr r (1 - r); then e-threads sum over types
__device__ __forceinline__ void mix_bwd_final_body(const float* __restrict__ part, int nblk, int M, int E, const float* __restrict__ r,
                                                   const float* __restrict__ types, int emb, float* __restrict__ dfcw, float* __restrict__ dfcb, int accumulate) {
    __shared__ float dz[MIX_MAX_TYPES][8];
    const int m = threadIdx.x >> 3, e = threadIdx.x & 7;
    if (m < M && e < E) {
        double s_ = 0.0;
        for (int b = 0; b < nblk; ++b) s_ += (double)part[((long long)m * nblk + b) * 8 + e];
        const float rr = r[m * E + e];
        dz[m][e] = (float)s_ * rr * (1.f - rr);
    }
    __syncthreads();
    if (m == 0 && e < E) {
        float sb = 0.f;
        for (int mm = 0; mm < M; ++mm) sb += dz[mm][e];
        dfcb[e] = accumulate ? dfcb[e] + sb : sb;
        for (int k = 0; k < emb; ++k) {
            float sw = 0.f;
            for (int mm = 0; mm < M; ++mm) sw += dz[mm][e] * types[mm * emb + k];
            dfcw[e * emb + k] = accumulate ? dfcw[e * emb + k] + sw : sw;
        }
    }
}

__global__ void mix_multi_bwd_final_kernel(const float* __restrict__ part, int nblk, int M, int E, const float* __restrict__ r,
                                           const float* __restrict__ types, int emb, float* __restrict__ dfcw, float* __restrict__ dfcb, int accumulate) {
    mix_bwd_final_body(part, nblk, M, E, r, types, emb, dfcw, dfcb, accumulate);
}

// ---- every CondConv2d layer of the model in ONE launch (forward) / one launch pair (backward).  A step mixes ~90 filters and takes
// their gradients apart again: 90 + 2 x 90 launches of 10-20 us each, 4 % of the bf16 step.  A job = one layer (or one half of a fused
// gamma | beta pair); the table lives in device memory, the grid is the concatenation of the jobs' block ranges.  Same bodies, same
// block-to-element mapping, same summation order as the per-layer launches: bit-identical results.
struct MixJob {
    const float* W; const float* fcw; const float* fcb; float* r;                  // r: [M][E], written forward, read backward
    float* tck[MIX_MAX_TYPES]; float* tkc[MIX_MAX_TYPES]; __bf16* btck[MIX_MAX_TYPES]; __bf16* btkc[MIX_MAX_TYPES];
    float* dW; float* dfcw; float* dfcb; float* part;                              // gradient sinks, partial-sum workspace [M][nblk][8]
    long long tap_tkc;
    int E, Co, Ci, T, ld_tck, ld_dw, block0, nblk, accumulate, ci_pitch;          // ci_pitch >= Ci: channel pitch of outputs and gradients
};
__device__ __forceinline__ const MixJob* mix_find_job(const MixJob* __restrict__ jobs, int njobs, int bx) {
    int lo = 0, hi = njobs - 1;                       // last job with block0 <= bx
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= bx) lo = mid; else hi = mid - 1; }
    return jobs + lo;
}
__global__ void mix_jobs_fwd_kernel(const MixJob* __restrict__ jobs, int njobs, const float* __restrict__ types, int emb) {
    const MixJob* j = mix_find_job(jobs, njobs, (int)blockIdx.x);
    const int m = blockIdx.y;
    mix_fwd_body(j->W, j->fcw, j->fcb, types, emb, j->r, j->tck[m], j->tkc[m], j->btck[m], j->btkc[m], j->E, j->Co, j->Ci, j->T, j->ld_tck,
                 j->tap_tkc, m, (int)blockIdx.x - j->block0, j->nblk, j->ci_pitch);
}
// dw: [njobs][MIX_MAX_TYPES] gradient pointers of this step (null: that label's filter got no gradient)
__global__ __launch_bounds__(256) void mix_jobs_bwd_kernel(const MixJob* __restrict__ jobs, int njobs, const float* const* __restrict__ dw, int M) {
    const MixJob* j = mix_find_job(jobs, njobs, (int)blockIdx.x);
    MixCPtrs g;
    const float* const* gp = dw + (long long)(j - jobs) * MIX_MAX_TYPES;
#pragma unroll
    for (int mm = 0; mm < MIX_MAX_TYPES; ++mm) g.p[mm] = mm < M ? gp[mm] : nullptr;
    mix_bwd_body(g, j->W, j->r, j->dW, j->part, M, j->E, j->Co, j->Ci, j->T, j->accumulate, j->ld_dw, (int)blockIdx.y,
                 (int)blockIdx.x - j->block0, j->nblk, j->ci_pitch);
}
__global__ void mix_jobs_bwd_final_kernel(const MixJob* __restrict__ jobs, const float* __restrict__ types, int emb, int M) {
    const MixJob* j = jobs + blockIdx.x;
    mix_bwd_final_body(j->part, j->nblk, M, j->E, j->r, types, emb, j->dfcw, j->dfcb, j->accumulate);
}

static int mix_blocks(long long total) { int b = mrdis_cdiv(total, 256); return b > 128 ? 128 : (b < 1 ? 1 : b); }

extern "C" int mrdis_mix_experts_fwd(const float* W, const float* r, float* w_tck, float* w_tkc,
                                     int E, int Co, int Ci, int T, void* stream) {
    if (!W || !r || E < 1 || E > 8 || Co < 1 || Ci < 1 || T < 1) return MRDIS_EINVAL;
    const long long total = (long long)Co * Ci * T;
    MRDIS_LAUNCH(mix_fwd_kernel, dim3(mix_blocks(total)), dim3(256), 0, (hipStream_t)stream, W, r, w_tck, w_tkc, E, Co, Ci, T);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" size_t mrdis_mix_experts_bwd_workspace(int E, int Co, int Ci, int T) {
    (void)E;
    return sizeof(float) * 8 * (size_t)mix_blocks((long long)Co * Ci * T);
}

extern "C" int mrdis_mix_experts_bwd(const float* dw_tck, const float* W, const float* r,
                                     float* dW, float* dr, void* workspace, size_t workspace_bytes,
                                     int E, int Co, int Ci, int T, void* stream) {
    if (!dw_tck || !W || !r || !dW || !dr || !workspace || E < 1 || E > 8) return MRDIS_EINVAL;
    const long long total = (long long)Co * Ci * T;
    const int nb = mix_blocks(total);
    if (workspace_bytes < sizeof(float) * 8 * (size_t)nb) return MRDIS_EWORKSPACE;
    MRDIS_LAUNCH(mix_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dw_tck, W, r, dW,
                       reinterpret_cast<float*>(workspace), E, Co, Ci, T);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(mix_bwd_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(workspace), nb, E, dr);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_mix_experts_routed_fwd(const float* W, const float* fc_w, const float* fc_b, const float* type_row, int emb,
                                            float* r_out, float* w_tck, float* w_tkc, int E, int Co, int Ci, int T, void* stream) {
    if (!W || !fc_w || !fc_b || !type_row || !r_out || !w_tck || !w_tkc || E < 1 || E > 8 || emb < 1 || emb > 16) return MRDIS_EINVAL;
    const long long total = (long long)Co * Ci * T;
    MRDIS_LAUNCH(mix_routed_fwd_kernel, dim3(mix_blocks(total)), dim3(256), 0, (hipStream_t)stream, W, fc_w, fc_b, type_row, emb,
                       r_out, w_tck, w_tkc, E, Co, Ci, T);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_mix_experts_routed_bwd(const float* dw_tck, const float* W, const float* r, const float* type_row, int emb,
                                            float* dW, float* dfc_w, float* dfc_b, void* workspace, size_t workspace_bytes,
                                            int E, int Co, int Ci, int T, void* stream) {
    if (!dw_tck || !W || !r || !type_row || !dW || !dfc_w || !dfc_b || !workspace || E < 1 || E > 8 || emb < 1 || emb > 16) return MRDIS_EINVAL;
    const long long total = (long long)Co * Ci * T;
    const int nb = mix_blocks(total);
    if (workspace_bytes < sizeof(float) * 8 * (size_t)nb) return MRDIS_EWORKSPACE;
    MRDIS_LAUNCH(mix_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dw_tck, W, r, dW,
                       reinterpret_cast<float*>(workspace), E, Co, Ci, T);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(mix_routed_bwd_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(workspace), nb, E, r, type_row, emb, dfc_w, dfc_b);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_mix_experts_routed_multi_fwd(const float* W, const float* fc_w, const float* fc_b, const float* types, int emb, int M,
                                                  float* r_out, float* const* w_tck, float* const* w_tkc,
                                                  void* const* w_bf16_tck, void* const* w_bf16_tkc, int ld_tck, long long tap_tkc,
                                                  int E, int Co, int Ci, int T, void* stream) {
    if (!W || !fc_w || !fc_b || !types || !r_out || !w_tck || !w_tkc || E < 1 || E > 8 || emb < 1 || emb > 16 || M < 1 || M > MIX_MAX_TYPES)
        return MRDIS_EINVAL;
    MixPtrs out{};
    if ((w_bf16_tck == nullptr) != (w_bf16_tkc == nullptr)) return MRDIS_EINVAL;
    for (int m = 0; m < M; ++m) {
        if (!w_tck[m] || !w_tkc[m]) return MRDIS_EINVAL;
        out.tck[m] = w_tck[m]; out.tkc[m] = w_tkc[m];
        if (w_bf16_tck != nullptr) {
            if (!w_bf16_tck[m] || !w_bf16_tkc[m]) return MRDIS_EINVAL;
            out.btck[m] = reinterpret_cast<__bf16*>(w_bf16_tck[m]); out.btkc[m] = reinterpret_cast<__bf16*>(w_bf16_tkc[m]);
        }
    }
    const long long total = (long long)Co * Ci * T;
    if (ld_tck <= 0) ld_tck = Co;
    if (tap_tkc <= 0) tap_tkc = (long long)Co * Ci;
    if (ld_tck < Co || tap_tkc < (long long)Co * Ci) return MRDIS_EINVAL;
    MRDIS_LAUNCH(mix_routed_multi_fwd_kernel, dim3(mix_blocks(total), M), dim3(256), 0, (hipStream_t)stream, W, fc_w, fc_b, types, emb,
                       r_out, out, E, Co, Ci, T, ld_tck, tap_tkc);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" size_t mrdis_mix_experts_routed_multi_bwd_workspace(int M, int E, int Co, int Ci, int T) {
    (void)E;
    return sizeof(float) * 8 * (size_t)mix_blocks((long long)Co * Ci * T) * (size_t)(M > 0 ? M : 1);
}

extern "C" int mrdis_mix_experts_routed_multi_bwd(const float* const* dw_tck, const float* W, const float* r, const float* types,
                                                  int emb, int M, float* dW, float* dfc_w, float* dfc_b, int accumulate, int ld_dw,
                                                  void* workspace, size_t workspace_bytes, int E, int Co, int Ci, int T, void* stream) {
    if (!dw_tck || !W || !r || !types || !dW || !dfc_w || !dfc_b || !workspace || E < 1 || E > 8 || emb < 1 || emb > 16 || M < 1 || M > MIX_MAX_TYPES)
        return MRDIS_EINVAL;
    const long long total = (long long)Co * Ci * T;
    const int nb = mix_blocks(total);
    if (workspace_bytes < mrdis_mix_experts_routed_multi_bwd_workspace(M, E, Co, Ci, T)) return MRDIS_EWORKSPACE;
    MixCPtrs dw{};
    for (int m = 0; m < M; ++m) dw.p[m] = dw_tck[m];
    MRDIS_LAUNCH(mix_multi_bwd_kernel, dim3(nb, M), dim3(256), 0, (hipStream_t)stream, dw, W, r, dW,
                       reinterpret_cast<float*>(workspace), M, E, Co, Ci, T, accumulate ? 1 : 0, ld_dw > 0 ? ld_dw : Co);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(mix_multi_bwd_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(workspace), nb, M, E, r, types, emb, dfc_w, dfc_b, accumulate ? 1 : 0);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" size_t mrdis_mix_job_bytes(void) { return sizeof(MixJob); }
extern "C" int mrdis_mix_job_blocks(int Co, int Ci, int T) { return mix_blocks((long long)Co * Ci * T); }

extern "C" int mrdis_mix_jobs_fwd(const void* jobs, int njobs, int total_blocks, const float* types, int emb, int M, void* stream) {
    if (!jobs || !types || njobs < 1 || total_blocks < njobs || emb < 1 || emb > 16 || M < 1 || M > MIX_MAX_TYPES) return MRDIS_EINVAL;
    MRDIS_LAUNCH(mix_jobs_fwd_kernel, dim3(total_blocks, M), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const MixJob*>(jobs), njobs, types, emb);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_mix_jobs_bwd(const void* jobs, int njobs, int total_blocks, const void* dw_table, const float* types, int emb, int M, void* stream) {
    if (!jobs || !dw_table || !types || njobs < 1 || total_blocks < njobs || emb < 1 || emb > 16 || M < 1 || M > MIX_MAX_TYPES) return MRDIS_EINVAL;
    MRDIS_LAUNCH(mix_jobs_bwd_kernel, dim3(total_blocks, M), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const MixJob*>(jobs), njobs,
                       reinterpret_cast<const float* const*>(dw_table), M);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(mix_jobs_bwd_final_kernel, dim3(njobs), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const MixJob*>(jobs), types, emb, M);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
