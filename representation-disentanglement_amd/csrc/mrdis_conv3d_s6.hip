// mrdis_conv3d_s6.hip -- 3x3x3 / stride 1 / pad 1 convolution with 16 input and 16 output channels, fp32 in / fp32 out, NDHWC (gfx950):
// the BasicBlock convolutions of the 3-D nets at full resolution (reference src/model.py:1856-1874: Conv3d(16, 16, 3, padding = 1) twice per
// block; NVNet3D on 4 x 4 x 128^3 runs six of them forward and six as data gradients per step -- 14 ms of a 68 ms step on the fp32 MFMA kernel
// conv3d16_kernel<16>, which is bound by the fp32 matrix pipe: 116 GFLOP per call at 98 TF/s).
//
// Six-product form (option split6, as mrdis_c16.hip / the 4 -> C kernel of mrdis_conv.hip): both fp32 operands are carried as three bf16 terms
// v = hi + mid + lo (each the bf16 rounding of what the terms before it left: 3 x 8 mantissa bits) and the six products of order <= 2
//     x_h w_h + x_m w_h + x_h w_m + x_l w_h + x_h w_l + x_m w_m
// are summed in fp32 on the bf16 matrix pipe; what is dropped is below 2^-24 of a product.  With K = 16 channels per tap two terms share one
// v_mfma_f32_16x16x32_bf16 (k-slots 0-15 = 16 channels of one term pair, 16-31 = of another): three MFMAs of 16 cycles per tap and 16 positions
//     A = [w_l | w_h], B = [x_h | x_l]      A = [w_m | w_m], B = [x_h | x_m]      A = [w_h | w_h], B = [x_h | x_m]          (smallest products first)
// instead of four fp32 MFMAs of 32 cycles: 2.67x less matrix-pipe time.
//
// Workgroup = 4 waves, persistent over 4 x 8 x 16-position boxes.  LDS: the halo'd input box (6 x 10 x 18 pixels) as [pixel][hi | mid | lo][16 ch]
// bf16 = 96 B per pixel, split ONCE per element on its way in (the next box's global loads are in flight in registers during a box's MFMAs), and
// the filter as [cube position][term, channel half][cout][8 ch] = 1.5 KB per tap, split once per workgroup.  A wave owns one depth slice of the box:
// eight rows of 16 positions that share every A operand -- per tap 3 + 16 ds_read_b128 for 24 MFMAs.  Both images are conflict-free for the lane
// groups a ds_read_b128 is served in ({0-3, 12-15, 20-27}, ...): a pixel's two 16-byte channel halves sit at 16-byte slots 6 i and 6 i + 1 (mod 16):
// the half-0 lanes of a group cover the even slots, the half-1 lanes the odd ones.
// D[cout][position]: lane = position, 4 registers = 4 consecutive couts: one 16-byte store per lane and row (+ bias, + residual: BasicBlock's x + y).
#include "mrdis_conv3d.h"

namespace {
typedef __bf16 s6_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s6_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int S6_TD = 4, S6_TH = 8, S6_TW = 16;
constexpr int S6_ID = S6_TD + 2, S6_IH = S6_TH + 2, S6_IW = S6_TW + 2;
constexpr int S6_NPX = S6_ID * S6_IH * S6_IW;          // 1080 pixels
constexpr int S6_PB = 96;                              // bytes per pixel
constexpr int S6_FT = 6 * 256;                         // bytes per cube position of the filter image
constexpr int S6_XR = (S6_NPX * 4 + 255) / 256;        // float4 staging items per thread (17)
constexpr size_t S6_LDS = (size_t)27 * S6_FT + (size_t)S6_NPX * S6_PB;     // 41,472 + 103,680 B

template <int V_> struct S6IC { static constexpr int value = V_; };
__device__ __forceinline__ void s6_split(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v; const float r1 = v - (float)h; m = (__bf16)r1; l = (__bf16)(r1 - (float)m);
}
}  // namespace

__global__ __launch_bounds__(256) void conv3d16_s6_kernel(const Conv3dParams p, int nboxes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s6_smem[];
    unsigned char* const fs = s6_smem;                        // filter image
    unsigned char* const xs = s6_smem + 27 * S6_FT;            // pixel image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4, half = kq & 1, up = kq >> 1;

    // ---- the filter, once per workgroup: cube position c = ((dd - dd_min) * 3 + (dh - dh_min)) * 3 + (dw - dw_min) of tap t (forward and flipped data-gradient
    //      tables alike), element (k = channel, j = cout) -> piece (term, k / 8), row j, slot k % 8
    for (int idx = tid; idx < 27 * 256; idx += 256) {
        const int t = idx >> 8, k = (idx >> 4) & 15, j = idx & 15;
        const int c = ((p.dd[t] - p.dd_min) * 3 + (p.dh[t] - p.dh_min)) * 3 + (p.dw[t] - p.dw_min);
        __bf16 h, m, l;
        s6_split(p.w[((long long)p.widx[t] * 16 + k) * 16 + j], h, m, l);
        __bf16* d = reinterpret_cast<__bf16*>(fs + c * S6_FT + (k >> 3) * 256 + j * 16) + (k & 7);
        d[0] = h; d[2 * 128] = m; d[4 * 128] = l;             // pieces (term, half) = term * 2 + half, 256 B = 128 bf16 each
    }
    // A operands: [w_h | w_h], [w_m | w_m], [w_l | w_h]: per-lane piece of each
    const int a1 = (0 * 2 + half) * 256 + l16 * 16, a2 = (1 * 2 + half) * 256 + l16 * 16, a3 = ((up ? 0 : 2) * 2 + half) * 256 + l16 * 16;
    // B operands of row g of this wave's depth slice at cube position (dz, dy, dx): pixel ((wave + dz) * IH + g + dy) * IW + l16 + dx
    //   B1 = [x_h | x_m], B3 = [x_h | x_l]
    const int b1 = ((wave * S6_IH) * S6_IW + l16) * S6_PB + up * 32 + half * 16, b3 = ((wave * S6_IH) * S6_IW + l16) * S6_PB + up * 64 + half * 16;

    // ---- staging roles (box-invariant): item = (pixel of the 6 x 10 x 18 block, channel quad)
    int s_l[S6_XR], s_c[S6_XR];
#pragma unroll
    for (int it = 0; it < S6_XR; ++it) {
        const int idx = tid + 256 * it, pi = idx >> 2, q = idx & 3;
        const int iz = pi / (S6_IH * S6_IW), rem = pi - iz * (S6_IH * S6_IW), iy = rem / S6_IW, ix = rem - iy * S6_IW;
        s_l[it] = idx < S6_NPX * 4 ? pi * S6_PB + q * 8 : -1;
        s_c[it] = (iz << 20) | (iy << 10) | ix;
    }
    const int qx = (tid & 3) * 4;                              // 256 % 4 == 0: the channel quad of every item of this thread
    float4 xr[S6_XR];
    auto load_box = [&](int box) {
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA; tt /= p.tilesA;
        const int tz = tt % p.tilesZ;
        const int n = tt / p.tilesZ;
        const int d_org = tz * S6_TD + p.dd_min, h_org = ta * S6_TH + p.dh_min, w_org = tb * S6_TW + p.dw_min;
        const float* __restrict__ in_n = p.in + (long long)n * p.Din * p.Hin * p.Win * p.ldin + qx;
#pragma unroll
        for (int it = 0; it < S6_XR; ++it) {
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int d = d_org + (s_c[it] >> 20), h = h_org + ((s_c[it] >> 10) & 1023), w_ = w_org + (s_c[it] & 1023);
            if (s_l[it] >= 0 && (unsigned)d < (unsigned)p.Din && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win)
                xr[it] = *reinterpret_cast<const float4*>(in_n + ((long long)(d * p.Hin + h) * p.Win + w_) * p.ldin);
        }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < S6_XR; ++it) {
            if (s_l[it] < 0) continue;
            const float xv[4] = {xr[it].x, xr[it].y, xr[it].z, xr[it].w};
            s6_bf16x4 hi, mid, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) { __bf16 h, m, l; s6_split(xv[c], h, m, l); hi[c] = h; mid[c] = m; lo[c] = l; }
            *reinterpret_cast<s6_bf16x4*>(xs + s_l[it]) = hi;
            *reinterpret_cast<s6_bf16x4*>(xs + s_l[it] + 32) = mid;
            *reinterpret_cast<s6_bf16x4*>(xs + s_l[it] + 64) = lo;
        }
    };

    const int co = 4 * kq;
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[r] = p.bias[co + r];
    }
    const bool vec_res = p.res != nullptr && (p.ldres % 4 == 0) && (((uintptr_t)p.res & 15) == 0);

    int box = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // neighbouring boxes (shared halos) on the same XCD / L2
    if (box < nboxes) load_box(box);
    store_box();
    __syncthreads();
    for (; box < nboxes; box += gridDim.x) {
        const int nxt = box + gridDim.x;
        if (nxt < nboxes) load_box(nxt);
        f32x4 acc[S6_TH];
#pragma unroll
        for (int g = 0; g < S6_TH; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        // software pipeline over the 27 cube positions: the 19 operand reads of position c + 1 are issued BETWEEN the 24 MFMAs of position c (one wave per SIMD:
        // nothing else hides the LDS latency -- without the interleave a position cost its read phase plus its MFMA phase, 25.7k cycles per box against 10.4k of MFMAs)
        s6_bf16x8 A[2][3], Bx[2][2][S6_TH];
        auto load_ops = [&](int c, auto SET_) {
            constexpr int set = decltype(SET_)::value;
            const int dz = c / 9, dy = (c / 3) % 3, dx = c % 3;
            const int po = ((dz * S6_IH + dy) * S6_IW + dx) * S6_PB;
            A[set][0] = *reinterpret_cast<const s6_bf16x8*>(fs + c * S6_FT + a1);
            A[set][1] = *reinterpret_cast<const s6_bf16x8*>(fs + c * S6_FT + a2);
            A[set][2] = *reinterpret_cast<const s6_bf16x8*>(fs + c * S6_FT + a3);
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) {
                Bx[set][0][g] = *reinterpret_cast<const s6_bf16x8*>(xs + b1 + po + g * (S6_IW * S6_PB));
                Bx[set][1][g] = *reinterpret_cast<const s6_bf16x8*>(xs + b3 + po + g * (S6_IW * S6_PB));
            }
        };
        auto mfmas = [&](auto SET_) {
            constexpr int set = decltype(SET_)::value;
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[set][2], Bx[set][1][g], acc[g], 0, 0, 0);      // w_l x_h + w_h x_l
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[set][1], Bx[set][0][g], acc[g], 0, 0, 0);      // w_m x_h + w_m x_m
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[set][0], Bx[set][0][g], acc[g], 0, 0, 0);      // w_h x_h + w_h x_m
        };
        load_ops(0, S6IC<0>{});
#pragma unroll
        for (int c = 0; c < 27; c += 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < 27) load_ops(c + 1, S6IC<1>{});
            mfmas(S6IC<0>{});
            if (c + 1 < 27) {
#pragma unroll
                for (int i = 0; i < 19; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < 27) load_ops(c + 2, S6IC<0>{});
                mfmas(S6IC<1>{});
                if (c + 2 < 27) {
#pragma unroll
                    for (int i = 0; i < 19; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // epilogue: D col = lane & 15 (position along W), rows 4 * (lane >> 4) + r (couts)
            int tt = box;
            const int tb = tt % p.tilesB; tt /= p.tilesB;
            const int ta = tt % p.tilesA; tt /= p.tilesA;
            const int tz = tt % p.tilesZ;
            const int n = tt / p.tilesZ;
            const int z = tz * S6_TD + wave, b = tb * S6_TW + l16;
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) {
                const int a = ta * S6_TH + g;
                if (z >= p.Z || a >= p.A || b >= p.B) continue;
                const long long po = ((long long)(n * p.Dout + z) * p.Hout + a) * p.Wout + b;
                float4 v = make_float4(acc[g][0] + bq[0], acc[g][1] + bq[1], acc[g][2] + bq[2], acc[g][3] + bq[3]);
                if (p.res != nullptr) {
                    const float* rsd = p.res + po * p.ldres + co;
                    if (vec_res) { const float4 rq = *reinterpret_cast<const float4*>(rsd); v.x += rq.x; v.y += rq.y; v.z += rq.z; v.w += rq.w; }
                    else { v.x += rsd[0]; v.y += rsd[1]; v.z += rsd[2]; v.w += rsd[3]; }
                }
                *reinterpret_cast<float4*>(p.out + po * p.ldout + co) = v;
            }
        }
        if (nxt < nboxes) { __syncthreads(); store_box(); __syncthreads(); }
    }
}

// p: the tap table as run_tapconv3d built it (dd_min / dh_min / dw_min set).  Takes: 16 -> 16 channels, all 27 taps of a stride-1 cube, 16-byte aligned
// NDHWC views -- the forward and the stride-1 data gradient of the BasicBlock convolutions.  Option split6: 1 (default) and 8 (this kernel only) select it.
int mrdis_run_conv3d16_s6(const Conv3dParams& p_in, long long ptiles_hint, hipStream_t s) {
    (void)ptiles_hint;
    const long long s6 = mrdis_opt(MRDIS_OPT_SPLIT6);
    if (s6 != 1 && s6 != 8) return MRDIS_EUNSUPPORTED;
    if (p_in.Cin != 16 || p_in.Cout != 16 || p_in.ntaps != 27 || p_in.is != 1 || p_in.os != 1 || p_in.od0 || p_in.oh0 || p_in.ow0) return MRDIS_EUNSUPPORTED;
    if (!p_in.vec_in || p_in.ldout % 4 != 0 || (((uintptr_t)p_in.out) & 15) != 0 || (((uintptr_t)p_in.w) & 3) != 0) return MRDIS_EUNSUPPORTED;
    bool seen[27] = {false};
    for (int t = 0; t < 27; ++t) {
        const int dz = p_in.dd[t] - p_in.dd_min, dy = p_in.dh[t] - p_in.dh_min, dx = p_in.dw[t] - p_in.dw_min;
        if (dz < 0 || dz > 2 || dy < 0 || dy > 2 || dx < 0 || dx > 2) return MRDIS_EUNSUPPORTED;
        seen[(dz * 3 + dy) * 3 + dx] = true;
    }
    for (int c = 0; c < 27; ++c) if (!seen[c]) return MRDIS_EUNSUPPORTED;
    Conv3dParams p = p_in;
    p.TD = S6_TD; p.TH = S6_TH; p.TW = S6_TW; p.TinD = S6_ID; p.TinH = S6_IH; p.TinW = S6_IW;
    p.tilesZ = mrdis_cdiv(p.Z, S6_TD); p.tilesA = mrdis_cdiv(p.A, S6_TH); p.tilesB = mrdis_cdiv(p.B, S6_TW);
    const long long nboxes = (long long)p.N * p.tilesZ * p.tilesA * p.tilesB;
    // small volumes: the generic kernels' 128-position boxes waste fewer positions and fill the chip better
    if (nboxes > 0x7fffffffLL || nboxes < 512 || (long long)p.Z * p.A * p.B < 32768) return MRDIS_EUNSUPPORTED;
    static int ncu = 0;
    if (!ncu) {
        if (hipFuncSetAttribute((const void*)conv3d16_s6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S6_LDS) != hipSuccess) return MRDIS_ELAUNCH;
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int nblk = nboxes < ncu ? (int)nboxes : ncu;          // one workgroup per CU (145 KB of LDS)
    mrdis_count(MRDIS_CNT_SPLIT6_C3D);
    MRDIS_LAUNCH(conv3d16_s6_kernel, dim3(nblk), dim3(256), S6_LDS, s, p, (int)nboxes);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// =========================================================================== weight gradient of the same layers
// dW[t][ci][co] = sum_q x[q][ci] dy[q - d_t][co] (d_t = tap offset; the tap shift sits on dy, so the x operand of a k-step is read once for the 27 taps) with
// both operands as three bf16 terms and six products per fp32 product, K = 32 POSITIONS per v_mfma_f32_16x16x32_bf16 (reference: the weight gradient of
// Conv3d(16, 16, 3, padding = 1), model.py:1861-1864; the fp32 kernel wgrad3d16_kernel<16> runs it at 82 TF/s: 920 us per call, 16 calls per NVNet3D step).
// Box = 4 x 8 x 16 positions of x (term planes [px][16 ch] bf16, rows padded to 18 pixels) + the halo'd 6 x 10 x 18 box of dy (term planes [px][16 co]).
// Positions are the k axis, so both operands come out of LDS through the transposing read ds_read_b64_tr_b16 (a 16-lane group addresses four 32-byte pixel rows
// and receives one channel / cout column of them).  A k-step = 32 positions = rows (ya, ya + 2) x 16 columns, k-block kq <-> row ya + 2 (kq & 1), columns
// 8 (kq >> 1) .. + 7: the two 16-lane groups a ds_read_b64_tr_b16 half serves together are then 2 x 576 B = 128 (mod 256) apart -- all 64 banks once.
// A wave owns one depth slice (four k-steps) and keeps the 16 x 16 accumulators of all 27 taps (108 registers); per k-step 6 + 27 x 6 transposing reads for
// 162 MFMAs, the dy reads of tap t + 1 between the MFMAs of tap t.  Slabs per workgroup (split-K over boxes) + the fp32 kernel's ordered reduction launch.
namespace {
typedef short s6_s16x4 __attribute__((ext_vector_type(4)));
constexpr int W6D_XROWP = 18 * 32;                         // bytes per x row (16 pixels + 2 of padding)
constexpr int W6D_XPLANE = S6_TD * S6_TH * W6D_XROWP;      // 18,432
constexpr int W6D_YPLANE = S6_NPX * 32;                    // 34,560
constexpr int W6D_XR = (S6_TD * S6_TH * S6_TW * 4) / 256;  // 8 float4 items of x per thread
constexpr int W6D_YR = S6_XR;                              // 17 of dy
constexpr size_t W6D_LDS = (size_t)3 * W6D_XPLANE + (size_t)3 * W6D_YPLANE;      // 158,976 B
static_assert(W6D_LDS >= (size_t)4 * 27 * 256 * 4, "the cross-wave reduction reuses the operand images");
}  // namespace

struct Wgrad3dS6Params {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, D, H, W, ldx, lddy;
    int tilesZ, tilesA, tilesB, numTiles, splits;
    int nCi, nCo;                      // 16-channel slices of x / dy: one workgroup column per (slice of x, slice of dy) pair, as wgrad3d16_kernel
};

__global__ __launch_bounds__(256) void wgrad3d16_s6_kernel(const Wgrad3dS6Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char w6_smem[];
    unsigned char* const xs = w6_smem;                         // [3 terms][32 rows][18 px][16 ch] bf16
    unsigned char* const ys = w6_smem + 3 * W6D_XPLANE;        // [3 terms][6][10][18 px][16 co] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int lb = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);      // the (cic, coc) workgroups of a split share x / dy boxes: same XCD
    const int coc = lb % p.nCo, cic = (lb / p.nCo) % p.nCi, split = lb / (p.nCo * p.nCi);
    const int q4 = l16 >> 2, p4 = l16 & 3, rowsel = kq & 1, xpos = 8 * (kq >> 1) + q4;
    // per-lane bases of the transposing reads (k-step row ya, tap (r, s, u) add compile-time offsets; the second read of an operand is + 4 pixels = 128 B)
    const int xbase = ((wave * S6_TH + 2 * rowsel) * 18 + xpos) * 32 + 8 * p4;
    const int ybase = ((wave * S6_IH + 2 * rowsel) * S6_IW + xpos) * 32 + 8 * p4;

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bs4[4] = {0.f, 0.f, 0.f, 0.f};                      // this thread's share of the bias gradient: couts 4 (tid & 3) .. + 3

    float4 xr[W6D_XR], yr[W6D_YR];
    const int qc = (tid & 3) * 4;
    auto load_box = [&](int box) {
        const bool on = box < p.numTiles;
        int tt = box;
        const int tb = tt % p.tilesB; tt /= p.tilesB;
        const int ta = tt % p.tilesA; tt /= p.tilesA;
        const int tz = tt % p.tilesZ;
        const int n = tt / p.tilesZ;
        const int z0 = tz * S6_TD, a0 = ta * S6_TH, b0 = tb * S6_TW;
        const float* __restrict__ xn = p.x + (long long)n * p.D * p.H * p.W * p.ldx + 16 * cic + qc;
        const float* __restrict__ yn = p.dy + (long long)n * p.D * p.H * p.W * p.lddy + 16 * coc + qc;
#pragma unroll
        for (int it = 0; it < W6D_XR; ++it) {
            const int px = (tid + 256 * it) >> 2;
            const int d = z0 + (px >> 7), h = a0 + ((px >> 4) & 7), w_ = b0 + (px & 15);
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on && d < p.D && h < p.H && w_ < p.W) xr[it] = *reinterpret_cast<const float4*>(xn + ((long long)(d * p.H + h) * p.W + w_) * p.ldx);
        }
#pragma unroll
        for (int it = 0; it < W6D_YR; ++it) {
            const int px = (tid + 256 * it) >> 2;
            const int iz = px / (S6_IH * S6_IW), rem = px - iz * (S6_IH * S6_IW), iy = rem / S6_IW, ix = rem - iy * S6_IW;
            const int d = z0 - 1 + iz, h = a0 - 1 + iy, w_ = b0 - 1 + ix;
            yr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on && px < S6_NPX && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                yr[it] = *reinterpret_cast<const float4*>(yn + ((long long)(d * p.H + h) * p.W + w_) * p.lddy);
        }
    };
    auto split4 = [](const float4& v, s6_bf16x4& hi, s6_bf16x4& mid, s6_bf16x4& lo) {
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { __bf16 h, m, l; s6_split(f[c], h, m, l); hi[c] = h; mid[c] = m; lo[c] = l; }
    };
    auto store_box = [&]() {
#pragma unroll
        for (int it = 0; it < W6D_XR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 2, q = idx & 3;
            s6_bf16x4 hi, mid, lo; split4(xr[it], hi, mid, lo);
            unsigned char* d = xs + ((px >> 4) * 18 + (px & 15)) * 32 + 8 * q;
            *reinterpret_cast<s6_bf16x4*>(d) = hi; *reinterpret_cast<s6_bf16x4*>(d + W6D_XPLANE) = mid; *reinterpret_cast<s6_bf16x4*>(d + 2 * W6D_XPLANE) = lo;
        }
#pragma unroll
        for (int it = 0; it < W6D_YR; ++it) {
            const int idx = tid + 256 * it, px = idx >> 2, q = idx & 3;
            if (px >= S6_NPX) continue;
            const int iz = px / (S6_IH * S6_IW), rem = px - iz * (S6_IH * S6_IW), iy = rem / S6_IW, ix = rem - iy * S6_IW;
            if (iz >= 1 && iz <= S6_TD && iy >= 1 && iy <= S6_TH && ix >= 1 && ix <= S6_TW) {      // the box's own positions: the bias gradient (fp32, before the split)
                bs4[0] += yr[it].x; bs4[1] += yr[it].y; bs4[2] += yr[it].z; bs4[3] += yr[it].w;
            }
            s6_bf16x4 hi, mid, lo; split4(yr[it], hi, mid, lo);
            unsigned char* d = ys + px * 32 + 8 * q;
            *reinterpret_cast<s6_bf16x4*>(d) = hi; *reinterpret_cast<s6_bf16x4*>(d + W6D_YPLANE) = mid; *reinterpret_cast<s6_bf16x4*>(d + 2 * W6D_YPLANE) = lo;
        }
    };
    auto tr_read = [](const unsigned char* a) -> s6_bf16x8 {
        union { s6_bf16x8 v; s6_s16x4 h[2]; } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s6_s16x4 __attribute__((address_space(3)))*)(a));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s6_s16x4 __attribute__((address_space(3)))*)(a + 4 * 32));
        return u.v;
    };

    int box = split;
    load_box(box);
    store_box();
    __syncthreads();
    for (; box < p.numTiles; box += p.splits) {
        load_box(box + p.splits);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ya = (ks >> 1) * 4 + (ks & 1);                  // row pairs (0, 2), (1, 3), (4, 6), (5, 7) of the wave's depth slice
            s6_bf16x8 ax[3], by[2][3];
#pragma unroll
            for (int term = 0; term < 3; ++term) ax[term] = tr_read(xs + term * W6D_XPLANE + xbase + ya * W6D_XROWP);
            auto load_by = [&](int t, auto SET_) {
                constexpr int set = decltype(SET_)::value;
                const int r = t / 9, s_ = (t / 3) % 3, u_ = t % 3;
                const int off = (((2 - r) * S6_IH + (ya + 2 - s_)) * S6_IW + (2 - u_)) * 32;
#pragma unroll
                for (int term = 0; term < 3; ++term) by[set][term] = tr_read(ys + term * W6D_YPLANE + ybase + off);
            };
            auto mfmas = [&](int t, auto SET_) {                      // six products of order <= 2 (terms 0 = hi, 1 = mid, 2 = lo), smallest first
                constexpr int set = decltype(SET_)::value;
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[1], by[set][1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[2], by[set][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], by[set][2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[1], by[set][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], by[set][1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[0], by[set][0], acc[t], 0, 0, 0);
            };
            load_by(0, S6IC<0>{});
#pragma unroll
            for (int t = 0; t < 27; t += 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 27) load_by(t + 1, S6IC<1>{});
                mfmas(t, S6IC<0>{});
                if (t + 1 < 27) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 2 < 27) load_by(t + 2, S6IC<0>{});
                    mfmas(t + 1, S6IC<1>{});
                    if (t + 2 < 27) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        store_box();
        __syncthreads();
    }
    // cross-wave reduction through LDS (fixed order), then slab[split][tap][16 ci][16 co] -- the layout wgrad3d16_reduce_kernel sums
    float* red = reinterpret_cast<float*>(w6_smem);            // [4 waves][27][256]
    float* out = p.slab + (long long)lb * 27 * 256;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 27 + t) * 256 + (4 * kq + r) * 16 + l16] = acc[t][r];
    __syncthreads();
    for (int i = tid; i < 27 * 256; i += 256)
        out[i] = (red[i] + red[27 * 256 + i]) + (red[2 * 27 * 256 + i] + red[3 * 27 * 256 + i]);
    if (p.bias_slab != nullptr && cic == 0) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[c * 256 + tid] = bs4[c];      // thread tid holds couts 4 (tid & 3) + c
        __syncthreads();
        if (tid < 16) {
            const int qd = tid >> 2, c = tid & 3;
            float t_ = 0.f;
            for (int k = qd; k < 256; k += 4) t_ += red[c * 256 + k];
            p.bias_slab[((long long)split * p.nCo + coc) * 16 + tid] = t_;
        }
    }
}

// Weight (+ bias) slabs of a 3x3x3 stride-1 layer whose channel counts are multiples of 16 (16 -> 16 at full resolution, 32 -> 32 one level down, 32 -> 16 of the
// VAE branch ...); *splits_out = the number of slabs per (ci slice, co slice) for wgrad3d16_reduce_kernel (CW = 16).  MRDIS_EUNSUPPORTED outside what the kernel
// covers (then the fp32 kernels run).  Option split6: 1 (default) and 9 (this kernel only) select it.
int mrdis_run_wgrad3d16_s6(const float* x, int ldx, const float* dy, int lddy, float* slab, size_t slab_bytes, int want_bias,
                           int N, int D, int H, int W, int Ci, int Co, int* splits_out, float** bias_slab_out, hipStream_t s) {
    const long long s6 = mrdis_opt(MRDIS_OPT_SPLIT6);
    if (s6 != 1 && s6 != 9) return MRDIS_EUNSUPPORTED;
    if (Ci % 16 != 0 || Co % 16 != 0 || Ci > 64 || Co > 64) return MRDIS_EUNSUPPORTED;
    if (ldx % 4 != 0 || lddy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)slab)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    Wgrad3dS6Params p{};
    p.x = x; p.dy = dy; p.N = N; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.lddy = lddy;
    p.nCi = Ci / 16; p.nCo = Co / 16;
    p.tilesZ = mrdis_cdiv(D, S6_TD); p.tilesA = mrdis_cdiv(H, S6_TH); p.tilesB = mrdis_cdiv(W, S6_TW);
    const long long nt = (long long)N * p.tilesZ * p.tilesA * p.tilesB;
    if (nt > 0x7fffffffLL || nt < 512 || (long long)D * H * W < 32768 || (long long)D * H * W >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.numTiles = (int)nt;
    static int ncu = 0;
    if (!ncu) {
        if (hipFuncSetAttribute((const void*)wgrad3d16_s6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W6D_LDS) != hipSuccess) return MRDIS_ELAUNCH;
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int pairs = p.nCi * p.nCo;
    int splits = ncu / pairs; if (splits < 1) splits = 1;         // one workgroup per CU (155 KB of LDS): splits x slice pairs fill the chip once
    if (splits > p.numTiles) splits = p.numTiles;
    p.splits = splits;
    const size_t need = sizeof(float) * ((size_t)splits * pairs * 27 * 256 + (size_t)splits * p.nCo * 16);
    if (slab_bytes < need) return MRDIS_EUNSUPPORTED;
    p.slab = slab;
    p.bias_slab = want_bias ? slab + (size_t)splits * pairs * 27 * 256 : nullptr;
    mrdis_count(MRDIS_CNT_SPLIT6_W3D);
    MRDIS_LAUNCH(wgrad3d16_s6_kernel, dim3(splits * pairs), dim3(256), W6D_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    *splits_out = splits; *bias_slab_out = p.bias_slab;
    return MRDIS_OK;
}
