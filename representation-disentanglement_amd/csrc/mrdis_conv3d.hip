// mrdis_conv3d.hip -- NDHWC direct 3-D convolution for gfx950 (MI355X), fp32: the Conv3d layers of the reference's
// 3-D networks (src/model.py:1856-2060: BasicBlock, UNet3D, VAEBranch -- 3x3x3 stride 1 / 2, pad 1; the 1x1x1
// layers go through the 2-D kernels on the (N*D, H, W) view).
//
// Same tap-table formulation as mrdis_conv.hip with a depth axis:
//
//   out[n, z*os+od0, a*os+oh0, b*os+ow0, co] = bias[co] (+ res[...]) +
//        sum_t sum_ci in[n, z*is + dd[t], a*is + dh[t], b*is + dw[t], ci] * w[widx[t]][ci][co]
//
//   forward              : is = stride, os = 1, d = r - pad, w = [T][Ci][Co]
//   data gradient, s = 1 : is = 1, os = 1, d = pad - r, w = [T][Co][Ci]
//   data gradient, s = 2 : eight output-parity classes, os = 2
//
// Workgroup (4 waves) = 128 output positions (TD x TH x TW box of one sample) x BN couts.  Per KC-channel chunk the
// halo'd input box ([pixel][KC+1], odd stride -> conflict-free A reads) and the [27][KC][BN] filter slab are staged
// once in LDS and all taps run out of it with v_mfma_f32_32x32x2_f32 (A = filter, B = pixels -> D[cout][position],
// one 16-byte store per four couts).  27 taps amortise a staged pixel three times better than the 2-D kernel does,
// so KC = 8 (70 KB of LDS, two workgroups per CU) already keeps staging below 10 % of the MFMA time.
//
// Weight gradient: transposed product, M = (tap, ci) in 32-row sub-tiles, N = 32 couts, K = output positions,
// split-K slabs + fixed-order reduction (bit-reproducible).  Stride-2 layers run as eight parity classes of the
// input (one tap group per class; inside a class the input view is stride 1 with doubled pitches), so the staged
// halo is the size of the position box instead of eight times it.
#include "mrdis_common.h"
#include "mrdis_conv3d.h"
#include <stdlib.h>

template <int KC, int BN>
__global__ __launch_bounds__(256) void tapconv3d_kernel(const Conv3dParams p) {
    constexpr int BM = 128;
    constexpr int S = KC + 1;
    constexpr int WAVES_N = (BN == 32) ? 1 : 2;
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int MSUB = (BM / 32) / WAVES_M;
    constexpr int NSUB = (BN / 32) / WAVES_N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab_in = reinterpret_cast<int*>(smem);
    int* tab_out = tab_in + BM;
    int* tap_xoff = tab_in + 2 * BM;
    int* tap_widx = tab_in + 2 * BM + 32;
    float* ws = smem + T3_TAB_INTS;
    float* xs = ws + p.ntaps * KC * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    int bid = mrdis_xcd_remap(blockIdx.x, gridDim.x);
    const int cot = bid % p.coTiles;
    int tile = bid / p.coTiles;
    const int tb = tile % p.tilesB; tile /= p.tilesB;
    const int ta = tile % p.tilesA; tile /= p.tilesA;
    const int tz = tile % p.tilesZ;
    const int n = tile / p.tilesZ;
    const int z0 = tz * p.TD, a0 = ta * p.TH, b0 = tb * p.TW, co0 = cot * BN;

    if (tid < BM) {
        const int m = tid;
        const int npos = p.TD * p.TH * p.TW;
        int tin = 0, tout = -1;
        if (m < npos) {
            const int pz = m / (p.TH * p.TW);
            const int rem = m - pz * p.TH * p.TW;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = (((pz * p.is) * p.TinH + ty * p.is) * p.TinW + tx * p.is) * S;
            const int z = z0 + pz, a = a0 + ty, b = b0 + tx;
            if (z < p.Z && a < p.A && b < p.B)
                tout = ((n * p.Dout + z * p.os + p.od0) * p.Hout + a * p.os + p.oh0) * p.Wout + b * p.os + p.ow0;   // host: < 2^31 positions
        }
        tab_in[m] = tin;
        tab_out[m] = tout;
    } else if (tid >= 256 - 32) {
        const int t = tid - (256 - 32);
        if (t < p.ntaps) {
            tap_xoff[t] = (((p.dd[t] - p.dd_min) * p.TinH + (p.dh[t] - p.dh_min)) * p.TinW + (p.dw[t] - p.dw_min)) * S;
            tap_widx[t] = p.widx[t];
        }
    }
    __syncthreads();

    int abase[MSUB];
#pragma unroll
    for (int i = 0; i < MSUB; ++i) abase[i] = tab_in[(wave_m * MSUB + i) * 32 + (lane & 31)] + (lane >> 5);
    const int bbase = (lane >> 5) * BN + wave_n * NSUB * 32 + (lane & 31);

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int i = 0; i < MSUB; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int tinHW = p.TinH * p.TinW;
    const int npix_in = p.TinD * tinHW;
    const int d_org = z0 * p.is + p.dd_min, h_org = a0 * p.is + p.dh_min, w_org = b0 * p.is + p.dw_min;
    const float* __restrict__ in_n = p.in + (long long)n * p.Din * p.Hin * p.Win * p.ldin;

    // one KC-chunk of MFMAs, every tap out of LDS; operands of the next group of k-steps are read before the MFMAs of
    // the current group issue (register double buffer, as in tapconv_kernel)
    constexpr int MPS = MSUB * NSUB;
    constexpr int G0 = MPS >= 4 ? 1 : (MPS == 2 ? 2 : 4);
    constexpr int G = (KC / 2 < G0) ? KC / 2 : G0;
    constexpr int NG = (KC / 2) / G;
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

    for (int c0 = 0; c0 < p.Cin; c0 += KC) {
        if (c0) __syncthreads();
        // ---- stage the halo'd input box: xs[pixel][KC+1]
        if (p.vec_in) {
            constexpr int Q = KC / 4;
            for (int idx = tid; idx < npix_in * Q; idx += 256) {
                const int pi = idx / Q, q = idx - pi * Q;
                const int iz = pi / tinHW;
                const int rem = pi - iz * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int d = d_org + iz, h = h_org + iy, w_ = w_org + ix, c = c0 + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)d < (unsigned)p.Din && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win && c < p.Cin)
                    v = *reinterpret_cast<const float4*>(in_n + ((long long)(d * p.Hin + h) * p.Win + w_) * p.ldin + c);
                float* dst = xs + pi * S + 4 * q;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < npix_in * KC; idx += 256) {
                const int pi = idx / KC, k = idx - pi * KC;
                const int iz = pi / tinHW;
                const int rem = pi - iz * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int d = d_org + iz, h = h_org + iy, w_ = w_org + ix, c = c0 + k;
                float v = 0.f;
                if ((unsigned)d < (unsigned)p.Din && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win && c < p.Cin)
                    v = in_n[((long long)(d * p.Hin + h) * p.Win + w_) * p.ldin + c];
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

    // ---- epilogue: D is [cout][position]; a lane owns one position per M sub-tile and groups of four consecutive couts
    const int half = lane >> 5;
    const bool vec_out = (p.ldout % 4 == 0) && (((uintptr_t)p.out & 15) == 0);
    const bool vec_res = p.res != nullptr && (p.ldres % 4 == 0) && (((uintptr_t)p.res & 15) == 0);
    const bool vec_bias = p.bias != nullptr && (((uintptr_t)p.bias & 15) == 0);
#pragma unroll
    for (int i = 0; i < MSUB; ++i) {
        const int po = tab_out[(wave_m * MSUB + i) * 32 + (lane & 31)];
        if (po < 0) continue;
        float* dst = p.out + (long long)po * p.ldout;
        const float* rsd = p.res ? p.res + (long long)po * p.ldres : nullptr;
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
                if (rsd != nullptr) {
                    if (vec_res && full) {
                        const float4 rq = *reinterpret_cast<const float4*>(rsd + co);
                        v.x += rq.x; v.y += rq.y; v.z += rq.z; v.w += rq.w;
                    } else {
                        v.x += rsd[co];
                        if (co + 1 < p.Cout) v.y += rsd[co + 1];
                        if (co + 2 < p.Cout) v.z += rsd[co + 2];
                        if (co + 3 < p.Cout) v.w += rsd[co + 3];
                    }
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

// ------------------------------------------------------------------ host side
struct Box3 { int TD, TH, TW; };

// best-utilisation box of <= 128 positions; ties prefer wide rows (coalesced staging), then tall boxes (halo ratio)
static Box3 choose_box(int Z, int A, int B) {
    Box3 best{1, 1, 1};
    double best_u = -1.0;
    long long best_halo = 0;
    for (int tw = 1; tw <= 32 && tw <= B; ++tw)
        for (int th = 1; th * tw <= 128 && th <= A; ++th) {
            int td = 128 / (tw * th); if (td > Z) td = Z; if (td < 1) td = 1;
            const double u = ((double)B / ((double)mrdis_cdiv(B, tw) * tw)) * ((double)A / ((double)mrdis_cdiv(A, th) * th)) *
                             ((double)Z / ((double)mrdis_cdiv(Z, td) * td)) * ((double)(tw * th * td) / 128.0);
            const long long halo = (long long)(td + 2) * (th + 2) * (tw + 2);
            const bool tie = u > best_u - 1e-9;
            if (u > best_u + 1e-9 || (tie && halo < best_halo) || (tie && halo == best_halo && tw > best.TW)) {
                best_u = u; best = {td, th, tw}; best_halo = halo;
            }
        }
    return best;
}

static size_t tapconv3d_lds(const Conv3dParams& p, int KC, int BN) {
    return sizeof(float) * ((size_t)T3_TAB_INTS + (size_t)p.ntaps * KC * BN + (size_t)p.TinD * p.TinH * p.TinW * (KC + 1));
}

template <int KC, int BN>
static int launch_tapconv3d_t(const Conv3dParams& p, size_t lds, int nblk, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {      // > 64 KB of dynamic LDS needs the opt-in
        if (hipFuncSetAttribute((const void*)tapconv3d_kernel<KC, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    MRDIS_LAUNCH((tapconv3d_kernel<KC, BN>), dim3(nblk), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ---------------------------------------------------------------- narrow-output variant (Cout <= 16, Cin <= 32)
// The full-resolution levels of the 3-D nets have 16 channels (BasicBlock 16 -> 16 six times per step at 128^3, conv1a
// 4 -> 16, vconv1 32 -> 16, the data gradient of ds1 32 -> 16): a 32-wide cout tile is half padding there.  This variant
// uses v_mfma_f32_16x16x4_f32 (A = filter [16 couts x 4 ch], B = pixels [4 ch x 16 positions]) and is PERSISTENT: the
// whole [27][Cin][16] filter is staged in LDS once per workgroup, then the workgroup walks boxes of 128 positions with a
// grid stride -- all input channels of the halo'd box in one LDS tile -- while the global loads of the next box are in
// flight in registers.  Everything a thread needs per box (which pixels it stages, where its positions sit in the tile)
// is box-invariant and computed once; a box costs an origin add and three range checks per staged item.
template <int KC>
__global__ __launch_bounds__(256) void conv3d16_kernel(const Conv3dParams p, int nboxes) {
    constexpr int S = KC + 1, BN = 16, QX = KC / 4;
    constexpr int XR = (KC == 32) ? 12 : 6;            // float4 items per thread; host: npix_in * QX <= XR * 256
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tap_xoff = reinterpret_cast<int*>(smem);    // [32]
    float* ws = smem + 32;                           // [tap][KC][16]
    float* xs = ws + p.ntaps * KC * BN;              // [pixel][KC+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int tinHW = p.TinH * p.TinW, npix_in = p.TinD * tinHW;

    if (tid < p.ntaps)
        tap_xoff[tid] = (((p.dd[tid] - p.dd_min) * p.TinH + (p.dh[tid] - p.dh_min)) * p.TinW + (p.dw[tid] - p.dw_min)) * S;
    for (int idx = tid; idx < p.ntaps * KC * BN; idx += 256) {
        const int row = idx >> 4, j = idx & 15;
        const int t = row / KC, k = row - t * KC;
        float v = 0.f;
        if (k < p.Cin && j < p.Cout) v = p.w[((long long)p.widx[t] * p.Cin + k) * p.Cout + j];
        ws[idx] = v;
    }
    // box-invariant roles
    int bbase[2], pz_[2], ty_[2], tx_[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = (wave * 2 + i) * 16 + l16;
        const int thw = p.TH * p.TW;
        int pz = m / thw; const int rem = m - pz * thw;
        int ty = rem / p.TW, tx = rem - ty * p.TW;
        if (pz >= p.TD) { pz = 0; ty = 0; tx = 0; pz_[i] = 1 << 20; } else pz_[i] = pz;      // beyond the box: never stored
        ty_[i] = ty; tx_[i] = tx;
        bbase[i] = (((pz * p.is) * p.TinH + ty * p.is) * p.TinW + tx * p.is) * S + kq;
    }
    int xl[XR], xc[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 256;
        xl[it] = -1; xc[it] = 0;
        if (idx < npix_in * QX) {
            const int pi = idx / QX, q = idx - pi * QX;
            const int iz = pi / tinHW;
            const int rem = pi - iz * tinHW;
            const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
            xl[it] = pi * S + 4 * q;
            xc[it] = (iz << 20) | (iy << 10) | ix;
        }
    }
    const int qx = (tid % QX) * 4;                     // 256 % QX == 0: the channel quad of every item of this thread
    const bool q_ok = qx < p.Cin;
    const int abase = kq * BN + l16;
    const int co = 4 * kq;
    const bool vec_out = (p.ldout % 4 == 0) && (((uintptr_t)p.out & 15) == 0);
    const bool vec_res = p.res != nullptr && (p.ldres % 4 == 0) && (((uintptr_t)p.res & 15) == 0);
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (co + r < p.Cout) bq[r] = p.bias[co + r];
    }

    float4 xr[XR];
    auto load_box = [&](int box) {
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA; tt /= p.tilesA;
        const int tz = tt % p.tilesZ;
        const int n = tt / p.tilesZ;
        const int d_org = tz * p.TD * p.is + p.dd_min, h_org = ta * p.TH * p.is + p.dh_min, w_org = tb * p.TW * p.is + p.dw_min;
        const float* __restrict__ in_n = p.in + (long long)n * p.Din * p.Hin * p.Win * p.ldin + qx;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int d = d_org + (xc[it] >> 20), h = h_org + ((xc[it] >> 10) & 1023), w_ = w_org + (xc[it] & 1023);
            if (xl[it] >= 0 && q_ok && (unsigned)d < (unsigned)p.Din && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win)
                xr[it] = *reinterpret_cast<const float4*>(in_n + ((long long)(d * p.Hin + h) * p.Win + w_) * p.ldin);
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it)
            if (xl[it] >= 0) { float* d = xs + xl[it]; d[0] = xr[it].x; d[1] = xr[it].y; d[2] = xr[it].z; d[3] = xr[it].w; }
    };

    int box = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // neighbouring boxes (shared halos) on the same XCD / L2
    if (box < nboxes) load_box(box);
    store_box();
    __syncthreads();
    for (; box < nboxes; box += gridDim.x) {
        const int nxt = box + gridDim.x;
        if (nxt < nboxes) load_box(nxt);
        f32x4 acc[2];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
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
        // epilogue: D col = lane & 15 (position), rows 4 * (lane >> 4) + r (couts)
        {
            int tt = box;
            const int tb = tt % p.tilesB; tt /= p.tilesB;
            const int ta = tt % p.tilesA; tt /= p.tilesA;
            const int tz = tt % p.tilesZ;
            const int n = tt / p.tilesZ;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int z = tz * p.TD + pz_[i], a = ta * p.TH + ty_[i], b = tb * p.TW + tx_[i];
                if (z >= p.Z || a >= p.A || b >= p.B || co >= p.Cout) continue;
                const long long po = ((long long)(n * p.Dout + z * p.os + p.od0) * p.Hout + a * p.os + p.oh0) * p.Wout + b * p.os + p.ow0;
                float v[4] = {acc[i][0] + bq[0], acc[i][1] + bq[1], acc[i][2] + bq[2], acc[i][3] + bq[3]};
                const bool full = co + 3 < p.Cout;
                if (p.res != nullptr) {
                    const float* rsd = p.res + po * p.ldres + co;
                    if (vec_res && full) {
                        const float4 rq = *reinterpret_cast<const float4*>(rsd);
                        v[0] += rq.x; v[1] += rq.y; v[2] += rq.z; v[3] += rq.w;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (co + r < p.Cout) v[r] += rsd[r];
                    }
                }
                float* dst = p.out + po * p.ldout + co;
                if (vec_out && full) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (co + r < p.Cout) dst[r] = v[r];
                }
            }
        }
        if (nxt < nboxes) { __syncthreads(); store_box(); __syncthreads(); }
    }
}

template <int KC>
static int launch_conv3d16_t(const Conv3dParams& p, size_t lds, int nblk, int nboxes, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv3d16_kernel<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    MRDIS_LAUNCH((conv3d16_kernel<KC>), dim3(nblk), dim3(256), lds, s, p, nboxes);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

static int run_tapconv3d(Conv3dParams p, hipStream_t s) {
    if (p.ntaps < 1 || p.ntaps > T3_TAPS) return MRDIS_EUNSUPPORTED;
    if (p.Z <= 0 || p.A <= 0 || p.B <= 0 || p.N <= 0) return MRDIS_OK;
    int dmax[3] = {p.dd[0], p.dh[0], p.dw[0]};
    p.dd_min = p.dd[0]; p.dh_min = p.dh[0]; p.dw_min = p.dw[0];
    for (int t = 1; t < p.ntaps; ++t) {
        if (p.dd[t] < p.dd_min) p.dd_min = p.dd[t];
        if (p.dh[t] < p.dh_min) p.dh_min = p.dh[t];
        if (p.dw[t] < p.dw_min) p.dw_min = p.dw[t];
        if (p.dd[t] > dmax[0]) dmax[0] = p.dd[t];
        if (p.dh[t] > dmax[1]) dmax[1] = p.dh[t];
        if (p.dw[t] > dmax[2]) dmax[2] = p.dw[t];
    }
    // stride-2 forward boxes read (2T+1)^3 pixels: keep the box small enough for the LDS budget
    Box3 bx = choose_box(p.Z, p.A, p.B);
    p.TD = bx.TD; p.TH = bx.TH; p.TW = bx.TW;
    p.TinD = (p.TD - 1) * p.is + (dmax[0] - p.dd_min) + 1;
    p.TinH = (p.TH - 1) * p.is + (dmax[1] - p.dh_min) + 1;
    p.TinW = (p.TW - 1) * p.is + (dmax[2] - p.dw_min) + 1;
    p.tilesZ = mrdis_cdiv(p.Z, p.TD); p.tilesA = mrdis_cdiv(p.A, p.TH); p.tilesB = mrdis_cdiv(p.B, p.TW);
    const long long ptiles = (long long)p.N * p.tilesZ * p.tilesA * p.tilesB;
    if ((long long)p.N * p.Dout * p.Hout * p.Wout >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if ((long long)p.Din * p.Hin * p.Win >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    int BN = p.Cout <= 32 ? 32 : 64;
    if (BN == 64 && ptiles * mrdis_cdiv(p.Cout, 64) < 256) BN = 32;
    { const int v = (int)mrdis_opt(MRDIS_OPT_BN3); if (v == 32 || v == 64) BN = v; }
    p.vec_in = (p.Cin % 4 == 0) && (p.ldin % 4 == 0) && (((uintptr_t)p.in & 15) == 0);
    p.vec_w = (p.Cout % 4 == 0) && (((uintptr_t)p.w & 15) == 0);
    if (p.Cin == 16 && p.Cout == 16 && !mrdis_opt(MRDIS_OPT_NO16_3D)) {       // full-resolution BasicBlock layers: six bf16 products per fp32 product (mrdis_conv3d_s6.hip)
        const int rc6 = mrdis_run_conv3d16_s6(p, ptiles, s);
        if (rc6 != MRDIS_EUNSUPPORTED) return rc6;
    }
    if (p.Cout <= 16 && p.Cin <= 32 && p.vec_in && ptiles <= 0x7fffffffLL && !mrdis_opt(MRDIS_OPT_NO16_3D)) {
        const int KC16 = p.Cin <= 4 ? 4 : (p.Cin <= 8 ? 8 : (p.Cin <= 16 ? 16 : 32));
        const long long npix = (long long)p.TinD * p.TinH * p.TinW;
        const size_t lds16 = sizeof(float) * ((size_t)32 + (size_t)p.ntaps * KC16 * 16 + (size_t)npix * (KC16 + 1));
        const int xr = KC16 == 32 ? 12 : 6;
        if (npix * (KC16 / 4) <= (long long)xr * 256 && lds16 <= 128 * 1024 && p.TinD < 1024 && p.TinH < 1024 && p.TinW < 1024) {
            const int per_cu = lds16 <= 52 * 1024 ? 3 : (lds16 <= 80 * 1024 ? 2 : 1);
            long long nblk = 256LL * per_cu;
            if (nblk > ptiles) nblk = ptiles;
            switch (KC16) {
                case 4: return launch_conv3d16_t<4>(p, lds16, (int)nblk, (int)ptiles, s);
                case 8: return launch_conv3d16_t<8>(p, lds16, (int)nblk, (int)ptiles, s);
                case 16: return launch_conv3d16_t<16>(p, lds16, (int)nblk, (int)ptiles, s);
                default: return launch_conv3d16_t<32>(p, lds16, (int)nblk, (int)ptiles, s);
            }
        }
    }
    int KC = p.Cin <= 4 ? 4 : 8;
    { const int v = (int)mrdis_opt(MRDIS_OPT_KC3); if ((v == 4 || v == 8) && v < KC) KC = v; }
    const size_t LDS_MAX = 80 * 1024;        // two workgroups per CU
    while (tapconv3d_lds(p, KC, BN) > LDS_MAX && KC > 4) KC >>= 1;
    while (tapconv3d_lds(p, KC, BN) > LDS_MAX && BN > 32) BN >>= 1;
    if (tapconv3d_lds(p, KC, BN) > LDS_MAX) return MRDIS_EUNSUPPORTED;
    p.coTiles = mrdis_cdiv(p.Cout, BN);
    const long long nblk = ptiles * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    const size_t lds = tapconv3d_lds(p, KC, BN);
#define T3_CASE(kc, bn) if (KC == kc && BN == bn) return launch_tapconv3d_t<kc, bn>(p, lds, (int)nblk, s)
    T3_CASE(4, 32); T3_CASE(4, 64); T3_CASE(8, 32); T3_CASE(8, 64);
#undef T3_CASE
    return MRDIS_EUNSUPPORTED;
}

static int check_conv3d_geom(int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad, int* Do, int* Ho, int* Wo) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return MRDIS_EINVAL;
    if (k != 3 || pad != 1) return MRDIS_EUNSUPPORTED;           // model.py:1861-1864, 1969-1984: every 3-D conv is 3x3x3 / pad 1
    if (stride != 1 && stride != 2) return MRDIS_EUNSUPPORTED;
    *Do = (D + 2 * pad - k) / stride + 1;
    *Ho = (H + 2 * pad - k) / stride + 1;
    *Wo = (W + 2 * pad - k) / stride + 1;
    if (*Do <= 0 || *Ho <= 0 || *Wo <= 0) return MRDIS_EINVAL;
    return MRDIS_OK;
}

// hybrid Winograd path (mrdis_wino.hip): F(2x2,3x3) in (h, w), direct in depth
int mrdis_run_wino3d(const float* x, int ldx, const float* w, const float* bias, const float* res, int ldres, float* y, int ldy,
                     int N, int D, int H, int W, int Ci, int Co, int flip, hipStream_t s);
size_t mrdis_wino_wgrad3d_workspace(int N, int D, int H, int W, int Ci, int Co);
int mrdis_run_wino_wgrad3d(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                           size_t workspace_bytes, int N, int D, int H, int W, int Ci, int Co, hipStream_t s);
static bool wino3d_wgrad_wanted(int N, int D, int H, int W, int Ci, int Co, int stride) {
    const int mode = (int)mrdis_opt(MRDIS_OPT_WINO);          // MRDIS_WINO at load; mrdis_set_option("wino", v) afterwards
    if (mode == 0 || stride != 1 || mrdis_wino_wgrad3d_workspace(N, D, H, W, Ci, Co) == 0) return false;
    if (mode == 2) return true;
    if (Co % 64 != 0 && Ci < 128) return false;
    return (long long)N * D * ((H + 3) / 4) * ((W + 7) / 8) >= 256;
}
// measured policy (tools/bench3d.py --layers): MRDIS_WINO = 0 never | 1 where it wins | 2 wherever it applies
static bool wino3d_wanted(int N, int D, int H, int W, int Ci, int Co, int stride) {
    const int mode = (int)mrdis_opt(MRDIS_OPT_WINO);          // MRDIS_WINO at load; mrdis_set_option("wino", v) afterwards
    if (mode == 0 || stride != 1) return false;
    if (mode == 2) return Ci >= 8 && Co >= 8;
    if (Ci < 16 || Co < 32) return false;
    const int cg = Co > 32 ? 2 : 1;
    const long long nblk = (long long)N * D * mrdis_cdiv((H + 1) / 2, 8) * mrdis_cdiv((W + 1) / 2, 8) * mrdis_cdiv(Co, 32 * cg);
    return nblk >= (cg == 2 ? 256 : 512);
}

extern "C" int mrdis_conv3d_fwd(const float* x, int ldx, const float* w_tck, const float* bias, const float* residual, int ldres,
                                float* y, int ldy, int N, int D, int H, int W, int Ci, int Co,
                                int k, int stride, int pad, void* stream) {
    int Do, Ho, Wo;
    int rc = check_conv3d_geom(N, D, H, W, Ci, Co, k, stride, pad, &Do, &Ho, &Wo);
    if (rc) return rc;
    if (!x || !w_tck || !y || ldx < Ci || ldy < Co || (residual && ldres < Co)) return MRDIS_EINVAL;
    if (wino3d_wanted(N, D, H, W, Ci, Co, stride)) {
        rc = mrdis_run_wino3d(x, ldx, w_tck, bias, residual, ldres, y, ldy, N, D, H, W, Ci, Co, 0, (hipStream_t)stream);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    Conv3dParams p{};
    p.in = x; p.w = w_tck; p.bias = bias; p.res = residual; p.out = y;
    p.N = N; p.Din = D; p.Hin = H; p.Win = W; p.Cin = Ci; p.ldin = ldx;
    p.Dout = Do; p.Hout = Ho; p.Wout = Wo; p.Cout = Co; p.ldout = ldy; p.ldres = ldres;
    p.Z = Do; p.A = Ho; p.B = Wo; p.os = 1; p.is = stride;
    p.ntaps = k * k * k;
    for (int r = 0; r < k; ++r)
        for (int s_ = 0; s_ < k; ++s_)
            for (int q = 0; q < k; ++q) {
                const int t = (r * k + s_) * k + q;
                p.dd[t] = r - pad; p.dh[t] = s_ - pad; p.dw[t] = q - pad; p.widx[t] = t;
            }
    return run_tapconv3d(p, (hipStream_t)stream);
}

extern "C" int mrdis_conv3d_bwd_data(const float* dy, int lddy, const float* w_tkc, float* dx, int lddx,
                                     int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad, void* stream) {
    int Do, Ho, Wo;
    int rc = check_conv3d_geom(N, D, H, W, Ci, Co, k, stride, pad, &Do, &Ho, &Wo);
    if (rc) return rc;
    if (!dy || !w_tkc || !dx || lddy < Co || lddx < Ci) return MRDIS_EINVAL;
    Conv3dParams base{};
    base.in = dy; base.w = w_tkc; base.out = dx;
    base.N = N; base.Din = Do; base.Hin = Ho; base.Win = Wo; base.Cin = Co; base.ldin = lddy;
    base.Dout = D; base.Hout = H; base.Wout = W; base.Cout = Ci; base.ldout = lddx;
    base.is = 1;
    if (stride == 1) {
        if (wino3d_wanted(N, D, H, W, Co, Ci, stride)) {
            rc = mrdis_run_wino3d(dy, lddy, w_tkc, nullptr, nullptr, 0, dx, lddx, N, D, H, W, Co, Ci, 1, (hipStream_t)stream);
            if (rc != MRDIS_EUNSUPPORTED) return rc;
        }
        Conv3dParams p = base;
        p.Z = D; p.A = H; p.B = W; p.os = 1;
        p.ntaps = k * k * k;
        for (int r = 0; r < k; ++r)
            for (int s_ = 0; s_ < k; ++s_)
                for (int q = 0; q < k; ++q) {
                    const int t = (r * k + s_) * k + q;
                    p.dd[t] = pad - r; p.dh[t] = pad - s_; p.dw[t] = pad - q; p.widx[t] = t;
                }
        return run_tapconv3d(p, (hipStream_t)stream);
    }
    // stride 2: dx[i] gathers dy[(i + pad - r) / 2] for the taps r with (i + pad - r) even
    for (int pd = 0; pd < 2; ++pd)
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                Conv3dParams p = base;
                p.Z = (D - pd + 1) / 2; p.A = (H - ph + 1) / 2; p.B = (W - pw + 1) / 2;
                p.os = 2; p.od0 = pd; p.oh0 = ph; p.ow0 = pw;
                p.ntaps = 0;
                for (int r = 0; r < k; ++r) {
                    if (((pd + pad - r) & 1) != 0) continue;
                    for (int s_ = 0; s_ < k; ++s_) {
                        if (((ph + pad - s_) & 1) != 0) continue;
                        for (int q = 0; q < k; ++q) {
                            if (((pw + pad - q) & 1) != 0) continue;
                            const int t = p.ntaps++;
                            p.dd[t] = (pd + pad - r) >> 1; p.dh[t] = (ph + pad - s_) >> 1; p.dw[t] = (pw + pad - q) >> 1;
                            p.widx[t] = (r * k + s_) * k + q;
                        }
                    }
                }
                if (p.ntaps == 0) return MRDIS_EUNSUPPORTED;
                rc = run_tapconv3d(p, (hipStream_t)stream);
                if (rc) return rc;
            }
    return MRDIS_OK;
}

// =========================================================================== weight gradient
#define W3_MAX_SLOTS 64
#define W3_MAX_GROUPS 8
struct Wgrad3dParams {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, Ci, Co, lddy;
    int Z, A, B;                  // dy extents
    int nslots;                   // nG * J * TPS; slot_ok marks the real taps
    int dd[W3_MAX_SLOTS], dh[W3_MAX_SLOTS], dw[W3_MAX_SLOTS];
    unsigned long long slot_ok;
    // per tap group: the input view it reads (stride 2: one parity class per group, doubled pitches)
    long long g_off[W3_MAX_GROUPS];
    int g_Din[W3_MAX_GROUPS], g_Hin[W3_MAX_GROUPS], g_Win[W3_MAX_GROUPS];
    int g_ddmin[W3_MAX_GROUPS], g_dhmin[W3_MAX_GROUPS], g_dwmin[W3_MAX_GROUPS];
    long long x_img, x_plane; int x_row, x_pix;      // element pitches of the input view
    int TD, TH, TW, TinD, TinH, TinW, tilesZ, tilesA, tilesB, numTiles;
    int CW, TPS, nCi, nCo, nG, base, splits, vec_x, vec_dy;
};

template <int J>
__global__ __launch_bounds__(256, 2) void wgrad3d_kernel(const Wgrad3dParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int* tab_in = reinterpret_cast<int*>(smem);
    int* tab_pos = tab_in + 128;
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
    const int gDin = p.g_Din[g], gHin = p.g_Hin[g], gWin = p.g_Win[g];
    const int g_dd_min = p.g_ddmin[g], g_dh_min = p.g_dhmin[g], g_dw_min = p.g_dwmin[g];
    const int c_lo = cic * 32, co_lo = coc * 32;

    const int tinHW = p.TinH * p.TinW, npix_in = p.TinD * tinHW, thw = p.TH * p.TW, npos = p.TD * thw;
    if (tid < 128) {
        const int m = tid;
        int tin = 0;
        if (m < npos) {
            const int pz = m / thw;
            const int rem = m - pz * thw;
            const int ty = rem / p.TW, tx = rem - ty * p.TW;
            tin = ((pz * p.TinH + ty) * p.TinW + tx) * S;
        }
        tab_in[m] = tin;
    }
    int loff[J];
    unsigned lvalid = 0;
    {
        const int tl = e / p.CW, cl = e - tl * p.CW;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int slot = (g * J + j) * p.TPS + tl;
            const bool ok = slot < p.nslots && ((p.slot_ok >> slot) & 1ull) && (c_lo + cl) < p.Ci;
            loff[j] = ok ? (((p.dd[slot] - g_dd_min) * p.TinH + (p.dh[slot] - g_dh_min)) * p.TinW + (p.dw[slot] - g_dw_min)) * S + cl : 0;
            lvalid |= (ok ? 1u : 0u) << j;
        }
    }
    __syncthreads();

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
        const int ta = tt % p.tilesA; tt /= p.tilesA;
        const int tz = tt % p.tilesZ;
        const int n = tt / p.tilesZ;
        const int z0 = tz * p.TD, a0 = ta * p.TH, b0 = tb * p.TW;
        const int d_org = z0 + g_dd_min, h_org = a0 + g_dh_min, w_org = b0 + g_dw_min;
        const float* __restrict__ xn = xg + (long long)n * p.x_img;
        __syncthreads();
        if (tid < 128) {
            const int m = tid;
            int pos = -1;
            if (m < npos) {
                const int pz = m / thw;
                const int rem = m - pz * thw;
                const int ty = rem / p.TW, tx = rem - ty * p.TW;
                const int z = z0 + pz, a = a0 + ty, bb = b0 + tx;
                if (z < p.Z && a < p.A && bb < p.B) pos = ((n * p.Z + z) * p.A + a) * p.B + bb;
            }
            tab_pos[m] = pos;
        }
        if (p.vec_x) {
            const int Q = p.CW >> 2;
            for (int idx = tid; idx < npix_in * Q; idx += 256) {
                const int pi = idx / Q, q = idx - pi * Q;
                const int iz = pi / tinHW;
                const int rem = pi - iz * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int d = d_org + iz, h = h_org + iy, w_ = w_org + ix, c = c_lo + 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)d < (unsigned)gDin && (unsigned)h < (unsigned)gHin && (unsigned)w_ < (unsigned)gWin && c < p.Ci)
                    v = *reinterpret_cast<const float4*>(xn + (long long)d * p.x_plane + (long long)h * p.x_row + (long long)w_ * p.x_pix + c);
                float* dst = xs + pi * S + 4 * q;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
        } else {
            for (int idx = tid; idx < npix_in * p.CW; idx += 256) {
                const int pi = idx / p.CW, k = idx - pi * p.CW;
                const int iz = pi / tinHW;
                const int rem = pi - iz * tinHW;
                const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
                const int d = d_org + iz, h = h_org + iy, w_ = w_org + ix, c = c_lo + k;
                float v = 0.f;
                if ((unsigned)d < (unsigned)gDin && (unsigned)h < (unsigned)gHin && (unsigned)w_ < (unsigned)gWin && c < p.Ci)
                    v = xn[(long long)d * p.x_plane + (long long)h * p.x_row + (long long)w_ * p.x_pix + c];
                xs[pi * S + k] = v;
            }
        }
        __syncthreads();   // tab_pos visible
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
    }
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

struct Wgrad3dTapMap { int slot[T3_TAPS]; };
__global__ void wgrad3d_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Ci, int Co,
                                      int CW, int TPS, int J, int nCi, int nCo, int base, int nslab, Wgrad3dTapMap map,
                                      const float* __restrict__ bslab, float* __restrict__ dbias) {
    __shared__ float red[16][65];
    const int total = T3_TAPS * Ci * Co;
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
        for (int k = y; k < nslab; k += SL, src += step) s_ += *src;
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
        else dbias[i - total] = t;
    }
}

struct Wgrad3dPlan {
    Wgrad3dParams p;
    Wgrad3dTapMap map;
    int J;
    size_t lds;
    long long slab_floats, bias_floats;
};

static int plan_wgrad3d(Wgrad3dPlan& pl, int N, int D, int H, int W, int ldx, int Ci, int Co, int k, int stride, int pad) {
    int Do, Ho, Wo;
    int rc = check_conv3d_geom(N, D, H, W, Ci, Co, k, stride, pad, &Do, &Ho, &Wo);
    if (rc) return rc;
    Wgrad3dParams& p = pl.p;
    p = Wgrad3dParams{};
    p.N = N; p.Ci = Ci; p.Co = Co; p.Z = Do; p.A = Ho; p.B = Wo;
    if ((long long)N * Do * Ho * Wo >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (Ci >= 32) { p.CW = 32; p.TPS = 1; p.nCi = mrdis_cdiv(Ci, 32); }
    else { p.CW = Ci <= 4 ? 4 : (Ci <= 8 ? 8 : (Ci <= 16 ? 16 : 32)); p.TPS = 32 / p.CW; p.nCi = 1; }
    p.nCo = mrdis_cdiv(Co, 32);
    int dmin[W3_MAX_GROUPS][3], dmax[W3_MAX_GROUPS][3];
    if (stride == 1) {
        // slots = taps in order; groups of J*TPS consecutive taps
        int J = p.TPS == 1 ? 9 : (p.TPS == 2 ? 7 : (p.TPS == 4 ? 7 : 4));
        pl.J = J;
        const int per = J * p.TPS;
        p.nG = mrdis_cdiv(T3_TAPS, per);
        p.nslots = p.nG * per;
        p.slot_ok = 0;
        for (int t = 0; t < T3_TAPS; ++t) {
            const int r = t / 9, s_ = (t / 3) % 3, q = t % 3;
            p.dd[t] = r - pad; p.dh[t] = s_ - pad; p.dw[t] = q - pad;
            p.slot_ok |= 1ull << t;
            pl.map.slot[t] = t;
        }
        for (int g = 0; g < p.nG; ++g) { p.g_off[g] = 0; p.g_Din[g] = D; p.g_Hin[g] = H; p.g_Win[g] = W; }
        p.x_pix = ldx; p.x_row = W * ldx; p.x_plane = (long long)H * W * ldx; p.x_img = (long long)D * H * W * ldx;
    } else {
        // one group per input parity class (pd, ph, pw); tap r reads x[2z + r - 1]: r = 1 -> even index z, r = 0 / 2 -> odd
        // index z - 1 / z.  Inside the class the view x_c[k] = x[2k + parity] is stride 1.
        const int per = 8;
        pl.J = per / p.TPS;
        if (pl.J < 1) return MRDIS_EUNSUPPORTED;
        p.nG = 8; p.nslots = 64; p.slot_ok = 0;
        for (int g = 0; g < 8; ++g) {
            const int pd = (g >> 2) & 1, ph = (g >> 1) & 1, pw = g & 1;      // 1 = odd class
            p.g_off[g] = ((long long)pd * H * W + (long long)ph * W + pw) * ldx;
            p.g_Din[g] = pd ? D / 2 : (D + 1) / 2; p.g_Hin[g] = ph ? H / 2 : (H + 1) / 2; p.g_Win[g] = pw ? W / 2 : (W + 1) / 2;
            int i = 0;
            for (int r = 0; r < 3; ++r) {
                if ((r != 1) != (pd == 1)) continue;
                for (int s_ = 0; s_ < 3; ++s_) {
                    if ((s_ != 1) != (ph == 1)) continue;
                    for (int q = 0; q < 3; ++q) {
                        if ((q != 1) != (pw == 1)) continue;
                        const int slot = g * per + i++;
                        p.dd[slot] = r == 0 ? -1 : 0; p.dh[slot] = s_ == 0 ? -1 : 0; p.dw[slot] = q == 0 ? -1 : 0;
                        p.slot_ok |= 1ull << slot;
                        pl.map.slot[(r * 3 + s_) * 3 + q] = slot;
                    }
                }
            }
        }
        p.x_pix = 2 * ldx; p.x_row = 2 * W * ldx; p.x_plane = 2LL * H * W * ldx; p.x_img = (long long)D * H * W * ldx;
    }
    const int per = pl.J * p.TPS;
    for (int g = 0; g < p.nG; ++g) {
        bool any = false;
        for (int i = 0; i < per; ++i) {
            const int s_ = g * per + i;
            if (s_ >= p.nslots || !((p.slot_ok >> s_) & 1ull)) continue;
            const int v[3] = {p.dd[s_], p.dh[s_], p.dw[s_]};
            for (int a = 0; a < 3; ++a) {
                if (!any || v[a] < dmin[g][a]) dmin[g][a] = v[a];
                if (!any || v[a] > dmax[g][a]) dmax[g][a] = v[a];
            }
            any = true;
        }
        if (!any) for (int a = 0; a < 3; ++a) dmin[g][a] = dmax[g][a] = 0;
        p.g_ddmin[g] = dmin[g][0]; p.g_dhmin[g] = dmin[g][1]; p.g_dwmin[g] = dmin[g][2];
    }
    const Box3 bx = choose_box(Do, Ho, Wo);
    p.TD = bx.TD; p.TH = bx.TH; p.TW = bx.TW;
    int ext[3] = {0, 0, 0};
    for (int g = 0; g < p.nG; ++g)
        for (int a = 0; a < 3; ++a) if (dmax[g][a] - dmin[g][a] > ext[a]) ext[a] = dmax[g][a] - dmin[g][a];
    p.TinD = p.TD + ext[0]; p.TinH = p.TH + ext[1]; p.TinW = p.TW + ext[2];
    p.tilesZ = mrdis_cdiv(Do, p.TD); p.tilesA = mrdis_cdiv(Ho, p.TH); p.tilesB = mrdis_cdiv(Wo, p.TW);
    const long long nt = (long long)N * p.tilesZ * p.tilesA * p.tilesB;
    if (nt > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.numTiles = (int)nt;
    p.base = p.nCi * p.nCo * p.nG;
    int splits = mrdis_cdiv(512, p.base);
    if (splits > p.numTiles) splits = p.numTiles;
    if (splits < 1) splits = 1;
    p.splits = splits;
    pl.lds = sizeof(float) * ((size_t)256 + 128 * 32 + (size_t)p.TinD * p.TinH * p.TinW * (p.CW + 1));
    if (pl.lds > 80 * 1024) return MRDIS_EUNSUPPORTED;
    pl.slab_floats = (long long)splits * p.base * pl.J * 1024;
    pl.bias_floats = (long long)splits * p.nCo * 32;
    return MRDIS_OK;
}

// ---------------------------------------------------------------- narrow weight gradient (Co <= 16, stride 1)
// With 16 couts a 32x32 MFMA tile is half empty; here the product runs on v_mfma_f32_16x16x4_f32 -- A = 16 rows
// (tap, ci) x 4 positions, B = 4 positions x 16 couts -- and a wave keeps the accumulators of ALL 27 taps (27 x 4
// registers for Ci = 16), so one B read (dy) feeds 27 MFMAs and every A read is one ds_read of the halo'd x box.  The
// waves of a workgroup split the 128 positions of a box; boxes are walked with a grid stride (split-K over workgroups)
// with the next box's x / dy in flight in registers.  Ci > 16 runs as 16-channel slices (one workgroup column each).
struct Wgrad3d16Params {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, D, H, W, Ci, ldx, Co, lddy;
    int TD, TH, TW, TinD, TinH, TinW, tilesZ, tilesA, tilesB, numTiles;
    int nCi, nCo, splits;
};

template <int CW>
__global__ __launch_bounds__(256) void wgrad3d16_kernel(const Wgrad3d16Params p) {
    constexpr int S = CW + 1, TPS = 16 / CW, NST = (T3_TAPS + TPS - 1) / TPS, QX = CW / 4;
    constexpr int XR = 6;                               // host: npix_in * QX <= 6 * 256
    constexpr int RS = CW == 16 ? 7 : (CW == 8 ? 5 : 3);   // sub-tiles per reduction round (host sizes the LDS for it)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dys = smem;                                  // [128][16]
    int* tab_in = reinterpret_cast<int*>(smem + 2048);  // [128] position -> offset of its pixel in the x tile
    float* xs = smem + 2048 + 128;                      // [pixel][CW+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int lb = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // the (cic, coc) workgroups of a split share x / dy boxes: same XCD
    const int coc = lb % p.nCo, cic = (lb / p.nCo) % p.nCi, split = lb / (p.nCo * p.nCi);
    const int c_lo = cic * 16, co_lo = coc * 16;
    const int tinHW = p.TinH * p.TinW, npix_in = p.TinD * tinHW, thw = p.TH * p.TW;

    // A rows: e = l16 -> (tap slot tl, channel cl)
    const int tl = l16 / CW, cl = l16 - tl * CW;
    int loff[NST];
    unsigned lvalid = 0;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int tap = st * TPS + tl;
        const bool ok = tap < T3_TAPS && (c_lo + cl) < p.Ci;
        const int r = tap / 9, s_ = (tap / 3) % 3, q = tap % 3;
        loff[st] = ok ? ((r * p.TinH + s_) * p.TinW + q) * S + cl : 0;
        lvalid |= (ok ? 1u : 0u) << st;
    }
    if (tid < 128) {
        const int m = tid;
        int pz = m / thw; const int rem = m - pz * thw;
        int ty = rem / p.TW, tx = rem - ty * p.TW;
        if (pz >= p.TD) { pz = 0; ty = 0; tx = 0; }         // dy row is zero there
        tab_in[m] = ((pz * p.TinH + ty) * p.TinW + tx) * S;
    }
    // staging roles (box-invariant)
    int xl[XR], xc[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 256;
        xl[it] = -1; xc[it] = 0;
        if (idx < npix_in * QX) {
            const int pi = idx / QX, q = idx - pi * QX;
            const int iz = pi / tinHW;
            const int rem = pi - iz * tinHW;
            const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
            xl[it] = pi * S + 4 * q;
            xc[it] = (iz << 20) | (iy << 10) | ix;
        }
    }
    const int qx = c_lo + (tid % QX) * 4;
    const bool qx_ok = qx < p.Ci;
    int yc[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int m = (tid + it * 256) >> 2;
        const int pz = m / thw; const int rem = m - pz * thw;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
        yc[it] = pz < p.TD ? ((pz << 20) | (ty << 10) | tx) : -1;
    }
    const int qy = co_lo + (tid & 3) * 4;
    const bool qy_ok = qy < p.Co;

    f32x4 acc[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) acc[st] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    float4 xr[XR], yr[2];
    auto load_box = [&](int box) {
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA; tt /= p.tilesA;
        const int tz = tt % p.tilesZ;
        const int n = tt / p.tilesZ;
        const int z0 = tz * p.TD, a0 = ta * p.TH, b0 = tb * p.TW;
        const float* __restrict__ xn = p.x + (long long)n * p.D * p.H * p.W * p.ldx + qx;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int d = z0 - 1 + (xc[it] >> 20), h = a0 - 1 + ((xc[it] >> 10) & 1023), w_ = b0 - 1 + (xc[it] & 1023);
            if (xl[it] >= 0 && qx_ok && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                xr[it] = *reinterpret_cast<const float4*>(xn + ((long long)(d * p.H + h) * p.W + w_) * p.ldx);
        }
        const float* __restrict__ dyn = p.dy + (long long)n * p.D * p.H * p.W * p.lddy + qy;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            yr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int z = z0 + (yc[it] >> 20), a = a0 + ((yc[it] >> 10) & 1023), b = b0 + (yc[it] & 1023);
            if (yc[it] >= 0 && qy_ok && z < p.D && a < p.H && b < p.W)
                yr[it] = *reinterpret_cast<const float4*>(dyn + ((long long)(z * p.H + a) * p.W + b) * p.lddy);
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it)
            if (xl[it] >= 0) { float* d = xs + xl[it]; d[0] = xr[it].x; d[1] = xr[it].y; d[2] = xr[it].z; d[3] = xr[it].w; }
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
#pragma unroll 1
        for (int g = 0; g < 8; ++g) {
            const int m = wave * 32 + g * 4 + kq;
            const float bv = dys[m * 16 + l16];
            bsum += bv;
            const float* xa = xs + tab_in[m];
            float av[NST];
#pragma unroll
            for (int st = 0; st < NST; ++st) av[st] = ((lvalid >> st) & 1u) ? xa[loff[st]] : 0.f;
#pragma unroll
            for (int st = 0; st < NST; ++st) acc[st] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st], bv, acc[st], 0, 0, 0);
        }
        if (nxt < p.numTiles) { __syncthreads(); store_box(); __syncthreads(); }
    }
    // cross-wave reduction through LDS in rounds of RS sub-tiles (fixed order), then slab[split][cic][st][16][16]
    float* red = smem;                                  // [4 waves][RS][256]
    float* out = p.slab + ((long long)lb * NST) * 256;
#pragma unroll
    for (int r0 = 0; r0 < NST; r0 += RS) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            if (r0 + j < NST) {
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * RS + j) * 256 + (4 * kq + r) * 16 + l16] = acc[r0 + j][r];
            }
        }
        __syncthreads();
        for (int i = tid; i < RS * 256; i += 256) {
            const int j = i >> 8;
            if (r0 + j < NST)
                out[(r0 + j) * 256 + (i & 255)] = (red[i] + red[RS * 256 + i]) + (red[2 * RS * 256 + i] + red[3 * RS * 256 + i]);
        }
    }
    if (p.bias_slab != nullptr && cic == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        __syncthreads();
        if (kq == 0) red[wave * 16 + l16] = bsum;
        __syncthreads();
        if (tid < 16) p.bias_slab[((long long)split * p.nCo + coc) * 16 + tid] = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
    }
}

__global__ void wgrad3d16_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Ci, int Co, int CW, int nCi, int nCo, int nslab,
                                        const float* __restrict__ bslab, float* __restrict__ dbias) {
    __shared__ float red[16][65];
    const int TPS = 16 / CW, NST = (T3_TAPS + TPS - 1) / TPS;
    const int total = T3_TAPS * Ci * Co;
    const int nout = total + (dbias != nullptr ? Co : 0);
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int SL = blockDim.y, y = threadIdx.y;
    float s_ = 0.f;
    if (i < total) {
        const int co = i % Co;
        const int r = i / Co;
        const int ci = r % Ci, t = r / Ci;
        const int cic = ci >> 4, cl = ci & 15;
        const int st = t / TPS, row = (t - st * TPS) * CW + cl;
        const long long stride = (long long)nCi * nCo * NST * 256;
        const float* src = slab + ((long long)(cic * nCo + (co >> 4)) * NST + st) * 256 + row * 16 + (co & 15) + (long long)y * stride;
        for (int k = y; k < nslab; k += SL, src += (long long)SL * stride) s_ += *src;
    } else if (i < nout) {
        const int co = i - total;
        for (int k = y; k < nslab; k += SL) s_ += bslab[(long long)k * (nCo * 16) + co];
    }
    red[y][threadIdx.x] = s_;
    __syncthreads();
    if (y == 0 && i < nout) {
        float t = 0.f;
        for (int k = 0; k < SL; ++k) t += red[k][threadIdx.x];
        if (i < total) dw[i] = t;
        else dbias[i - total] = t;
    }
}

struct Wgrad3d16Plan { Wgrad3d16Params p; int CW; size_t lds; long long slab_floats, bias_floats; bool ok; };

static void plan_wgrad3d16(Wgrad3d16Plan& pl, int N, int D, int H, int W, int ldx, int Ci, int Co, int stride) {
    pl.ok = false;
    if (stride != 1 || Co % 4 != 0 || (Co > 16 && Co % 16 != 0) || Ci % 4 != 0 || (Ci > 16 && Ci % 16 != 0) || mrdis_opt(MRDIS_OPT_NO16_3D)) return;
    Wgrad3d16Params& p = pl.p;
    p = Wgrad3d16Params{};
    p.N = N; p.D = D; p.H = H; p.W = W; p.Ci = Ci; p.ldx = ldx; p.Co = Co;
    pl.CW = Ci <= 4 ? 4 : (Ci <= 8 ? 8 : 16);
    p.nCi = Ci > 16 ? Ci / 16 : 1;
    p.nCo = Co > 16 ? Co / 16 : 1;
    const Box3 bx = choose_box(D, H, W);
    p.TD = bx.TD; p.TH = bx.TH; p.TW = bx.TW;
    p.TinD = p.TD + 2; p.TinH = p.TH + 2; p.TinW = p.TW + 2;
    const long long npix = (long long)p.TinD * p.TinH * p.TinW;
    if (npix * (pl.CW / 4) > 6 * 256 || p.TinD >= 1024 || p.TinH >= 1024 || p.TinW >= 1024) return;
    p.tilesZ = mrdis_cdiv(D, p.TD); p.tilesA = mrdis_cdiv(H, p.TH); p.tilesB = mrdis_cdiv(W, p.TW);
    const long long nt = (long long)N * p.tilesZ * p.tilesA * p.tilesB;
    if (nt > 0x7fffffffLL || (long long)D * H * W >= 0x7fffffffLL) return;
    p.numTiles = (int)nt;
    int splits = 512 / (p.nCi * p.nCo);
    if (splits > p.numTiles) splits = p.numTiles;
    if (splits < 1) splits = 1;
    p.splits = splits;
    const int TPS = 16 / pl.CW, NST = (T3_TAPS + TPS - 1) / TPS, RS = pl.CW == 16 ? 7 : (pl.CW == 8 ? 5 : 3);
    size_t fl = (size_t)2048 + 128 + (size_t)npix * (pl.CW + 1);
    if (fl < (size_t)4 * RS * 256) fl = (size_t)4 * RS * 256;
    pl.lds = sizeof(float) * fl;
    pl.slab_floats = (long long)splits * p.nCi * p.nCo * NST * 256;
    pl.bias_floats = (long long)splits * p.nCo * 16;
    pl.ok = true;
}

extern "C" size_t mrdis_conv3d_bwd_weight_workspace(int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad) {
    Wgrad3dPlan pl;
    if (plan_wgrad3d(pl, N, D, H, W, Ci, Ci, Co, k, stride, pad)) return 0;
    size_t need = sizeof(float) * (size_t)(pl.slab_floats + pl.bias_floats) + 256;
    Wgrad3d16Plan p16;
    plan_wgrad3d16(p16, N, D, H, W, Ci, Ci, Co, stride);
    if (p16.ok) { const size_t n16 = sizeof(float) * (size_t)(p16.slab_floats + p16.bias_floats) + 256; if (n16 > need) need = n16; }
    if (stride == 1) { const size_t nw = mrdis_wino_wgrad3d_workspace(N, D, H, W, Ci, Co); if (nw > need) need = nw; }
    return need;
}

template <int J>
static int launch_wgrad3d_t(const Wgrad3dPlan& pl, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wgrad3d_kernel<J>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    MRDIS_LAUNCH((wgrad3d_kernel<J>), dim3(pl.p.splits * pl.p.base), dim3(256), pl.lds, s, pl.p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_conv3d_bwd_weight(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias,
                                       void* workspace, size_t workspace_bytes,
                                       int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad, void* stream) {
    Wgrad3dPlan pl;
    int rc = plan_wgrad3d(pl, N, D, H, W, ldx, Ci, Co, k, stride, pad);
    if (rc) return rc;
    if (!x || !dy || !dw_tck || !workspace || ldx < Ci || lddy < Co) return MRDIS_EINVAL;
    if (workspace_bytes < sizeof(float) * (size_t)(pl.slab_floats + pl.bias_floats)) return MRDIS_EWORKSPACE;
    if (((uintptr_t)workspace & 15) != 0) return MRDIS_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (wino3d_wgrad_wanted(N, D, H, W, Ci, Co, stride)) {
        rc = mrdis_run_wino_wgrad3d(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, D, H, W, Ci, Co, s);
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    if (Ci % 16 == 0 && Co % 16 == 0 && k == 3 && stride == 1 && pad == 1 && !mrdis_opt(MRDIS_OPT_NO16_3D) &&
        !wino3d_wgrad_wanted(N, D, H, W, Ci, Co, stride)) {      // six bf16 products per fp32 product (mrdis_conv3d_s6.hip)
        int nsl = 0; float* bsl = nullptr;
        rc = mrdis_run_wgrad3d16_s6(x, ldx, dy, lddy, reinterpret_cast<float*>(workspace), workspace_bytes, dbias != nullptr, N, D, H, W, Ci, Co, &nsl, &bsl, s);
        if (rc == MRDIS_OK) {
            const long long nout6 = (long long)T3_TAPS * Ci * Co + (dbias ? Co : 0);
            int SL = 1;
            while (SL < 16 && SL * 8 <= nsl) SL <<= 1;
            MRDIS_LAUNCH(wgrad3d16_reduce_kernel, dim3(mrdis_cdiv(nout6, 64)), dim3(64, SL), 0, s, reinterpret_cast<float*>(workspace), dw_tck, Ci, Co, 16,
                               Ci / 16, Co / 16, nsl, bsl, dbias);
            MRDIS_CHECK_LAUNCH();
            return MRDIS_OK;
        }
        if (rc != MRDIS_EUNSUPPORTED) return rc;
    }
    {
        Wgrad3d16Plan p16;
        plan_wgrad3d16(p16, N, D, H, W, ldx, Ci, Co, stride);
        const bool al = (ldx % 4 == 0) && (lddy % 4 == 0) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0);
        if (p16.ok && al && workspace_bytes >= sizeof(float) * (size_t)(p16.slab_floats + p16.bias_floats)) {
            Wgrad3d16Params& q = p16.p;
            q.x = x; q.dy = dy; q.lddy = lddy;
            q.slab = reinterpret_cast<float*>(workspace);
            q.bias_slab = dbias ? q.slab + p16.slab_floats : nullptr;
            const int nblk = q.splits * q.nCi * q.nCo;
            if (p16.CW == 16) MRDIS_LAUNCH((wgrad3d16_kernel<16>), dim3(nblk), dim3(256), p16.lds, s, q);
            else if (p16.CW == 8) MRDIS_LAUNCH((wgrad3d16_kernel<8>), dim3(nblk), dim3(256), p16.lds, s, q);
            else MRDIS_LAUNCH((wgrad3d16_kernel<4>), dim3(nblk), dim3(256), p16.lds, s, q);
            MRDIS_CHECK_LAUNCH();
            const long long nout16 = (long long)T3_TAPS * Ci * Co + (dbias ? Co : 0);
            int SL = 1;
            while (SL < 16 && SL * 8 <= q.splits) SL <<= 1;
            MRDIS_LAUNCH(wgrad3d16_reduce_kernel, dim3(mrdis_cdiv(nout16, 64)), dim3(64, SL), 0, s, q.slab, dw_tck, Ci, Co, p16.CW,
                               q.nCi, q.nCo, q.splits, q.bias_slab, dbias);
            MRDIS_CHECK_LAUNCH();
            return MRDIS_OK;
        }
    }
    Wgrad3dParams& p = pl.p;
    p.x = x; p.dy = dy; p.lddy = lddy;
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + pl.slab_floats : nullptr;
    p.vec_x = (Ci % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0) && (p.CW % 4 == 0);
    p.vec_dy = (Co % 4 == 0) && (lddy % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    switch (pl.J) {
        case 1: rc = launch_wgrad3d_t<1>(pl, s); break;
        case 2: rc = launch_wgrad3d_t<2>(pl, s); break;
        case 4: rc = launch_wgrad3d_t<4>(pl, s); break;
        case 7: rc = launch_wgrad3d_t<7>(pl, s); break;
        case 8: rc = launch_wgrad3d_t<8>(pl, s); break;
        case 9: rc = launch_wgrad3d_t<9>(pl, s); break;
        default: return MRDIS_EUNSUPPORTED;
    }
    if (rc) return rc;
    const long long nout = (long long)T3_TAPS * Ci * Co + (dbias ? Co : 0);
    int SL = 1;
    while (SL < 16 && SL * 8 <= p.splits) SL <<= 1;
    MRDIS_LAUNCH(wgrad3d_reduce_kernel, dim3(mrdis_cdiv(nout, 64)), dim3(64, SL), 0, s, p.slab, dw_tck, Ci, Co,
                       p.CW, p.TPS, pl.J, p.nCi, p.nCo, p.base, p.splits, pl.map, p.bias_slab, dbias);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
