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
#pragma unroll
        for (int c = 0; c < 27; ++c) {
            const int dz = c / 9, dy = (c / 3) % 3, dx = c % 3;
            const int po = ((dz * S6_IH + dy) * S6_IW + dx) * S6_PB;
            const s6_bf16x8 A1 = *reinterpret_cast<const s6_bf16x8*>(fs + c * S6_FT + a1);
            const s6_bf16x8 A2 = *reinterpret_cast<const s6_bf16x8*>(fs + c * S6_FT + a2);
            const s6_bf16x8 A3 = *reinterpret_cast<const s6_bf16x8*>(fs + c * S6_FT + a3);
            s6_bf16x8 B1[S6_TH], B3[S6_TH];
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) {
                B1[g] = *reinterpret_cast<const s6_bf16x8*>(xs + b1 + po + g * (S6_IW * S6_PB));
                B3[g] = *reinterpret_cast<const s6_bf16x8*>(xs + b3 + po + g * (S6_IW * S6_PB));
            }
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3, B3[g], acc[g], 0, 0, 0);      // w_l x_h + w_h x_l
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1[g], acc[g], 0, 0, 0);      // w_m x_h + w_m x_m
#pragma unroll
            for (int g = 0; g < S6_TH; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1[g], acc[g], 0, 0, 0);      // w_h x_h + w_h x_m
        }
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
