// mrdis_s6conv.hip -- the tap-table convolution (mrdis_conv.hip: any filter extent, stride 1 | 2, forward and data gradient incl. the four parity
// classes of a stride-2 data gradient) with fp32-equivalent arithmetic on the bf16 matrix pipe: "six products" (option split6, as the thin 3x3 kernels
// of mrdis_c16.hip / mrdis_co4.hip).  Both fp32 operands are split into three bf16 terms v = h + m + l (each the RNE rounding of what the previous ones
// left: 24 significant bits in all) and the six products of order <= 2 -- w_h x_h, w_h x_m, w_m x_h, w_m x_m, w_h x_l, w_l x_h -- are summed in the fp32
// accumulator of v_mfma_f32_32x32x16_bf16; what is dropped (w_m x_l, w_l x_m, w_l x_l) is below 2^-24 of |w||x| per product.  Two products share one
// MFMA through its K axis: the 16 reduction slots of an instruction are 8 channels x 2 term pairs,
//     [w_l | w_h] x [x_h | x_l],   [w_m | w_m] x [x_h | x_m],   [w_h | w_h] x [x_h | x_m]
// i.e. three MFMAs per (tap, 8 channels, 32 couts x 32 positions) = 6/16 of the matrix-pipe time of the fp32 instruction (v_mfma_f32_16x16x4_f32 runs at
// 1/16 of the bf16 rate).  For the layers that have no Winograd form -- the 4x4 stride-2 encoder convolutions (model.py:2104 under :1935-1990) and the
// 3x3 stride-2 ones -- the fp32 tap kernel spends half its time in the matrix pipe at 45-115 TFLOP/s; here the pipe's share drops to a fifth and the
// kernel is bound by staging.
//
// The FILTER is split once per launch sequence by mrdis_s6_filter_image (below) into an image the kernel copies without arithmetic:
//     image[8-channel group q][32-cout group g][tap][cout e of 32][h | m | l][8 channels] bf16      (48 bytes per row, 1536 per (q, g, tap))
// (a workgroup re-splitting its filter chunk for every position tile put 60 % of the split arithmetic and eight 16-byte loads per thread on the critical
// path of ONE wave.)  The INPUT is split on its way into LDS: xs[pixel][q][h | m | l][8] bf16, pitch = an odd number of 16-byte slots, so the 16 lanes of a
// ds_read_b128 group land on distinct slots of the 256-byte bank row (48-byte filter rows: 3 slots).  A lane's MFMA operand is one ds_read_b128: lanes 0-31
// take the first term of the pair, lanes 32-63 the second.  Stride-2 inputs are staged with the columns de-interleaved by parity (a tap reads ONE parity:
// neighbouring positions are neighbouring slots again) and the row pitch of the pixel image is padded so that the tile rows a 32-lane half covers start on
// the slots the rule above leaves free.
//
// Structure (that of bconv_kernel, mrdis_bf16.hip): PERSISTENT workgroups of 4 waves walk (position tile, channel chunk) items; the global loads of item
// i + 1 (input pieces fp32, filter pieces bf16 terms, register-staged, issued unconditionally from clamped addresses) fly while item i is multiplied; the
// k-steps (tap, 8 channels) of an item are software-pipelined: the operand reads of step s + 1 are issued between the MFMAs of step s.
// D[cout][position]: a lane owns one position and 4-cout groups, i.e. 16-byte stores (NHWC).
#include "mrdis_tapconv.h"
#include "mrdis_s6conv.h"

namespace {
typedef __bf16 s6t_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s6t_bf16x2 __attribute__((ext_vector_type(2)));
typedef float s6t_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned s6t_u32x4 __attribute__((ext_vector_type(4)));
constexpr int S6T_ROW = 48, S6T_PLANE = 32 * S6T_ROW;      // bytes per filter row (8 channels x 3 terms) / per (q, g, tap)

__host__ __device__ constexpr int s6t_pitch(int kc) { return ((kc / 8) * 3) % 2 ? (kc / 8) * 48 : (kc / 8) * 48 + 16; }

// v -> (h, m, l) for 8 values, conversions in pairs (v_cvt_pk_bf16_f32 rounds two values per instruction)
__device__ __forceinline__ void s6t_split8(const float* v, s6t_u32x4& h, s6t_u32x4& m, s6t_u32x4& l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const s6t_f32x2 a = {v[2 * k], v[2 * k + 1]};
        const s6t_bf16x2 hh = __builtin_convertvector(a, s6t_bf16x2);
        const s6t_f32x2 r1 = a - __builtin_convertvector(hh, s6t_f32x2);
        const s6t_bf16x2 mm = __builtin_convertvector(r1, s6t_bf16x2);
        const s6t_f32x2 r2 = r1 - __builtin_convertvector(mm, s6t_f32x2);
        const s6t_bf16x2 ll = __builtin_convertvector(r2, s6t_bf16x2);
        h[k] = __builtin_bit_cast(unsigned, hh); m[k] = __builtin_bit_cast(unsigned, mm); l[k] = __builtin_bit_cast(unsigned, ll);
    }
}

template <int V_> struct S6TIC { static constexpr int value = V_; };

// XR / WR: register-staged input pieces (8 channels of a pixel, fp32) / filter pieces (16 bytes of the image) per thread (host: the tile fits them)
template <int KC, int WP, int WC, int XR, int WR>
__device__ __forceinline__ void s6conv_body(const TapConvParams& p, const S6ConvGeom& g, const int bx) {
    constexpr int PB = s6t_pitch(KC);              // bytes per pixel row of the input image
    constexpr int QX = KC / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char s6t_smem[];
    unsigned char* const ws = s6t_smem;            // [q][cout group j][tap][32][48]
    unsigned char* const xs = ws + QX * WC * p.ntaps * S6T_PLANE;
    __shared__ int sofs[MRDIS_MAX_TAPS * 4];       // per k-step (tap, 8-channel group): byte offset of the tap's pixel relative to the position's own slot (+ 48 per group)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, e = lane & 31;
    const int cot = bx % p.coTiles, wg = bx / p.coTiles;
    const int co0 = cot * 32 * WC;
    const int TinWp = g.TinWp;                     // padded row pitch of the pixel image (host: mrdis_run_s6conv)
    const int hw = (p.TinW + 1) >> 1;
    const int tinHW = p.TinH * p.TinW, npix_in = p.NB * tinHW, npos = p.NB * p.TH * p.TW;
    const int nsteps = p.ntaps * QX;
    auto xs_col = [&](int ix) { return p.is == 2 ? (ix & 1) * hw + (ix >> 1) : ix; };
    if (tid < nsteps) { const int t = tid / QX, ks = tid - t * QX; sofs[tid] = ((p.dh[t] - p.dh_min) * TinWp + xs_col(p.dw[t] - p.dw_min)) * PB + 48 * ks; }

    int abase[WP], pos_nb[WP], pos_ty[WP], pos_tx[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int m = 32 * (wave * WP + i) + e;
        int nb = 0, ty = 0, tx = 0;
        if (m < npos) { nb = m / (p.TH * p.TW); const int rem = m - nb * p.TH * p.TW; ty = rem / p.TW; tx = rem - ty * p.TW; }
        else nb = -1;
        pos_nb[i] = nb; pos_ty[i] = ty; pos_tx[i] = tx;
        abase[i] = (nb < 0) ? 0 : ((nb * p.TinH + ty * p.is) * TinWp + tx) * PB;
    }
    // staging descriptors (tile-invariant): input pieces (pixel, q), filter pieces (16 bytes of plane (q, j, tap))
    int x_desc[XR], x_dst[XR];
    const int nx = npix_in * QX;
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 256;
        x_desc[it] = -1; x_dst[it] = 0;
        if (idx < nx) {
            const int pi = idx / QX, q = idx - pi * QX;
            const int nb = pi / tinHW; const int rem = pi - nb * tinHW;
            const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
            x_desc[it] = (nb << 16) | (iy << 8) | ix;
            x_dst[it] = ((nb * p.TinH + iy) * TinWp + xs_col(ix)) * PB + q * 48;
        }
    }
    const int G = g.groups, T = g.taps_img;        // 32-cout groups and taps of the filter image
    const int nw = QX * WC * p.ntaps * (S6T_PLANE / 16);
    int w_src[WR];                                 // byte offset of the piece in the image at chunk 0 (host: the image is < 2^31 bytes), or -1 (no piece / couts past the image: zeros); its LDS offset is 16 (tid + 256 it)
#pragma unroll
    for (int it = 0; it < WR; ++it) {
        const int idx = tid + it * 256;
        w_src[it] = -1;
        if (idx < nw) {
            const int o = idx % (S6T_PLANE / 16), pl = idx / (S6T_PLANE / 16);      // plane (q, j, t) of the LDS image
            const int t = pl % p.ntaps, j = (pl / p.ntaps) % WC, q = pl / (p.ntaps * WC);
            const int gi = cot * WC + j;
            if (gi < G) w_src[it] = ((q * G + gi) * T + p.widx[t]) * S6T_PLANE + 16 * o;
        }
    }
    const unsigned char* const wimg = reinterpret_cast<const unsigned char*>(p.w_bf16);
    const long long w_chunk = (long long)QX * G * T * S6T_PLANE;      // bytes per channel chunk of the image
    float4 xr[XR][2];
    s6t_u32x4 wr[WR];
    unsigned x_ok = 0;                             // bit it: piece it holds image data (else zeros: padding, past the tile)
    auto tile_origin = [&](int tile, int& n0, int& a0, int& b0) {
        const int tb = tile % p.tilesB; tile /= p.tilesB;
        const int ta = tile % p.tilesA;
        n0 = (tile / p.tilesA) * p.NB; a0 = ta * p.TH; b0 = tb * p.TW;
    };
    // every load is issued unconditionally from a clamped address (a branch per piece makes hipcc drain the outstanding loads at every join: they would run one after the other)
    auto load_item = [&](int tile, int chunk, bool want_w) {
        int n0, a0, b0; tile_origin(tile, n0, a0, b0);
        const int h_org = a0 * p.is + p.dh_min, w_org = b0 * p.is + p.dw_min, c0 = chunk * KC;
        x_ok = 0;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int d = x_desc[it] < 0 ? 0 : x_desc[it];
            const int n = n0 + (d >> 16), h = h_org + ((d >> 8) & 255), w_ = w_org + (d & 255);
            const bool ok = x_desc[it] >= 0 && n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win;
            const long long off = ok ? ((long long)(n * p.Hin + h) * p.Win + w_) * p.ldin + c0 + 8 * ((tid + it * 256) % QX) : 0;
            xr[it][0] = *reinterpret_cast<const float4*>(p.in + off); xr[it][1] = *reinterpret_cast<const float4*>(p.in + off + 4);
            x_ok |= ok ? (1u << it) : 0u;
        }
        if (want_w) {
            const unsigned char* src = wimg + (long long)chunk * w_chunk;
#pragma unroll
            for (int it = 0; it < WR; ++it) wr[it] = *reinterpret_cast<const s6t_u32x4*>(src + (w_src[it] < 0 ? 0 : w_src[it]));
        }
    };
    auto store_item = [&](bool have_w) {
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            if (tid + it * 256 < nx) {
                const bool live = (x_ok >> it) & 1u;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = live ? (&xr[it][0].x)[k] : 0.f;
                s6t_u32x4 h, m, l;
                s6t_split8(v, h, m, l);
                unsigned char* d = xs + x_dst[it];
                *reinterpret_cast<s6t_u32x4*>(d) = h; *reinterpret_cast<s6t_u32x4*>(d + 16) = m; *reinterpret_cast<s6t_u32x4*>(d + 32) = l;
            }
        }
        if (have_w) {
#pragma unroll
            for (int it = 0; it < WR; ++it)
                if (tid + it * 256 < nw) *reinterpret_cast<s6t_u32x4*>(ws + 16 * (tid + it * 256)) = w_src[it] < 0 ? s6t_u32x4{0u, 0u, 0u, 0u} : wr[it];
        }
    };

    f32x16 acc[WC][WP];
    const bool w_resident = g.nchunks == 1;
    int tile = wg, chunk = 0;
    if (tile >= g.tiles) return;
#ifdef S6T_STAMPS      // diagnostic build (tools/build_abl.sh, tools/s6conv_stamps.py): s_memtime stamps of workgroups 0-3, per wave
    int n_stamp = 0;
    auto stamp = [&](int tag) {
        if (g.stamps != nullptr && bx < 4 && n_stamp < g.cap_stamps) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) g.stamps[((long long)(bx * 4 + wave)) * g.cap_stamps + n_stamp] = (t << 4) | (unsigned)tag;
            ++n_stamp;
        }
    };
#define S6T_STAMP(tag) stamp(tag)
#else
#define S6T_STAMP(tag)
#endif
    load_item(tile, 0, true);
    bool first = true;
    const bool lrelu = (p.epilogue & MRDIS_EPI_LRELU) != 0;
    // per-lane term of each operand: B1 = [x_h | x_l], B2 = [x_h | x_m]; A1 = [w_l | w_h], A2 = [w_m | w_m], A3 = [w_h | w_h]
    const int ob1 = half ? 32 : 0, ob2 = half ? 16 : 0, oa1 = half ? 0 : 32;
    const unsigned char* const wlane = ws + e * S6T_ROW;
    const int jstride = p.ntaps * S6T_PLANE, qstride = WC * p.ntaps * S6T_PLANE;
    // software pipeline over the k-steps: with one or two waves per SIMD nothing else hides the LDS latency (un-pipelined, a step cost its offset read + its
    // operand reads + its MFMAs one after the other, 3 x the MFMA time)
    constexpr int NRD = 2 * WP + 3 * WC, NMF = 3 * WP * WC;
    s6t_bf16x8 Bq[2][2][WP], Aq[2][3][WC];
    auto load_ops = [&](int s_, int xo, auto SET_) {
        constexpr int set = decltype(SET_)::value;
        const int t = s_ / QX, ks = s_ - t * QX;
        const unsigned char* wrow = wlane + ks * qstride + t * S6T_PLANE;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const unsigned char* px = xs + abase[i] + xo;
            Bq[set][0][i] = *reinterpret_cast<const s6t_bf16x8*>(px + ob1);
            Bq[set][1][i] = *reinterpret_cast<const s6t_bf16x8*>(px + ob2);
        }
#pragma unroll
        for (int j = 0; j < WC; ++j) {
            Aq[set][0][j] = *reinterpret_cast<const s6t_bf16x8*>(wrow + j * jstride + oa1);
            Aq[set][1][j] = *reinterpret_cast<const s6t_bf16x8*>(wrow + j * jstride + 16);
            Aq[set][2][j] = *reinterpret_cast<const s6t_bf16x8*>(wrow + j * jstride);
        }
    };
    auto mfmas = [&](auto SET_) {
        constexpr int set = decltype(SET_)::value;
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int i = 0; i < WP; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[set][0][j], Bq[set][0][i], acc[j][i], 0, 0, 0);      // w_l x_h + w_h x_l
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int i = 0; i < WP; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[set][1][j], Bq[set][1][i], acc[j][i], 0, 0, 0);      // w_m x_h + w_m x_m
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int i = 0; i < WP; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aq[set][2][j], Bq[set][1][i], acc[j][i], 0, 0, 0);      // w_h x_h + w_h x_m
    };
    auto interleave = [&]() {                      // one MFMA, then its share of the next step's reads
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            const int cnt = (NRD * (i + 1)) / NMF - (NRD * i) / NMF;
            if (cnt == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (cnt == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (cnt == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
    };
    while (tile < g.tiles) {
        S6T_STAMP(1);
        __syncthreads();                           // everyone is done reading the previous item's LDS images
        S6T_STAMP(2);
        store_item(first || !w_resident);
        S6T_STAMP(3);
        __syncthreads();
        S6T_STAMP(4);
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == g.nchunks) { nchunk = 0; ntile = tile + g.tile_stride; }
        if (ntile < g.tiles) load_item(ntile, nchunk, !w_resident);
        S6T_STAMP(5);
        first = false;
        if (chunk == 0) {
#pragma unroll
            for (int j = 0; j < WC; ++j)
#pragma unroll
                for (int i = 0; i < WP; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        }
        int oa = sofs[0], ob = sofs[nsteps > 1 ? 1 : 0];
        load_ops(0, oa, S6TIC<0>{});
        for (int s_ = 0; s_ < nsteps; s_ += 2) {
            __builtin_amdgcn_sched_barrier(0);
            oa = sofs[s_ + 2 < nsteps ? s_ + 2 : 0];
            if (s_ + 1 < nsteps) load_ops(s_ + 1, ob, S6TIC<1>{});
            mfmas(S6TIC<0>{});
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            ob = sofs[s_ + 3 < nsteps ? s_ + 3 : 0];
            if (s_ + 2 < nsteps) load_ops(s_ + 2, oa, S6TIC<0>{});
            if (s_ + 1 < nsteps) {
                mfmas(S6TIC<1>{});
                interleave();
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        S6T_STAMP(6);
        if (chunk == g.nchunks - 1) {
            // epilogue: D[cout][position]; a lane owns one position per block and the couts 8 q + 4 half .. + 3 (q = 0 .. 3)
            int n0, a0, b0; tile_origin(tile, n0, a0, b0);
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                if (pos_nb[i] < 0) continue;
                const int n = n0 + pos_nb[i], a = a0 + pos_ty[i], b = b0 + pos_tx[i];
                if (n >= p.N || a >= p.A || b >= p.B) continue;
                float* dst = p.out + ((long long)(n * p.Hout + a * p.os + p.oh0) * p.Wout + b * p.os + p.ow0) * p.ldout;
#pragma unroll
                for (int j = 0; j < WC; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = co0 + 32 * j + 8 * q + 4 * half;
                        if (co >= p.Cout) continue;
                        float4 v = make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
                        if (p.bias) { const float4 bb = *reinterpret_cast<const float4*>(p.bias + co); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                        if (lrelu) { v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y; v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w; }
                        *reinterpret_cast<float4*>(dst + co) = v;
                    }
            }
        }
        S6T_STAMP(7);
        tile = ntile; chunk = nchunk;
    }
}

template <int KC, int WP, int WC, int XR, int WR>
__global__ __launch_bounds__(256) void s6conv_kernel(const TapConvParams p, const S6ConvGeom g) { s6conv_body<KC, WP, WC, XR, WR>(p, g, (int)blockIdx.x); }

// the four output-parity classes of a stride-2 data gradient in one launch (blockIdx.y = class), as tapconv_pack_kernel
struct S6ConvPack { TapConvParams c[4]; S6ConvGeom g[4]; int grid[4]; };
template <int KC, int WP, int WC, int XR, int WR>
__global__ __launch_bounds__(256) void s6conv_pack_kernel(const S6ConvPack pk) {
    const int y = blockIdx.y;
    if ((int)blockIdx.x >= pk.grid[y]) return;
    s6conv_body<KC, WP, WC, XR, WR>(pk.c[y], pk.g[y], (int)blockIdx.x);
}

template <int KC, int WP, int WC, int XR, int WR>
int launch_s6conv_t(const TapConvParams& p, const S6ConvGeom& g, int grid, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)s6conv_kernel<KC, WP, WC, XR, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return MRDIS_ELAUNCH;
        attr_set = true;
    }
    MRDIS_LAUNCH((s6conv_kernel<KC, WP, WC, XR, WR>), dim3(grid), dim3(256), lds, s, p, g);
    MRDIS_CHECK_LAUNCH();
    mrdis_count(MRDIS_CNT_SPLIT6_TAP);
    return MRDIS_OK;
}
template <int KC, int WP, int WC, int XR, int WR>
int launch_s6conv_pack_t(const S6ConvLaunch (&L)[4], hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)s6conv_pack_kernel<KC, WP, WC, XR, WR>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) return MRDIS_ELAUNCH;
        attr_set = true;
    }
    S6ConvPack pk;
    int gx = 0; size_t lds = 0;
    for (int k = 0; k < 4; ++k) { pk.c[k] = L[k].p; pk.g[k] = L[k].g; pk.grid[k] = L[k].grid; if (L[k].grid > gx) gx = L[k].grid; if (L[k].lds > lds) lds = L[k].lds; }
    MRDIS_LAUNCH((s6conv_pack_kernel<KC, WP, WC, XR, WR>), dim3(gx, 4), dim3(256), lds, s, pk);
    MRDIS_CHECK_LAUNCH();
    mrdis_count(MRDIS_CNT_SPLIT6_TAP);
    return MRDIS_OK;
}

int s6t_ncu() {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

// workgroups of one instantiation a CU holds at a given dynamic LDS size (the runtime's answer: registers AND LDS), cached per (instantiation, form, size)
template <typename K>
int s6t_occ_of(K kernel, size_t lds) {
    static size_t seen_lds[4]; static int seen_n[4]; static int nseen = 0;
    for (int i = 0; i < nseen; ++i) if (seen_lds[i] == lds) return seen_n[i];
    int n = 0;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, lds) != hipSuccess || n < 1) n = 1;
    seen_lds[nseen & 3] = lds; seen_n[nseen & 3] = n; if (nseen < 4) ++nseen;
    return n;
}

// w [tap][reduction channel][output channel] fp32 -> the image above; one thread per row (q, g, tap, e)
__global__ void s6_filter_image_kernel(const float* __restrict__ w, int taps, int Cred, int Cout, unsigned char* __restrict__ img) {
    const int G = (Cout + 31) / 32, Q = Cred / 8;
    const long long rows = (long long)Q * G * taps * 32;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(r & 31); long long rr = r >> 5;
        const int t = (int)(rr % taps); rr /= taps;
        const int gi = (int)(rr % G), q = (int)(rr / G);
        const int co = 32 * gi + e;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = co < Cout ? w[((long long)t * Cred + 8 * q + k) * Cout + co] : 0.f;
        s6t_u32x4 h, m, l;
        s6t_split8(v, h, m, l);
        unsigned char* d = img + r * S6T_ROW;
        *reinterpret_cast<s6t_u32x4*>(d) = h; *reinterpret_cast<s6t_u32x4*>(d + 16) = m; *reinterpret_cast<s6t_u32x4*>(d + 32) = l;
    }
}
}  // namespace

extern "C" size_t mrdis_s6_filter_image_bytes(int taps, int Cred, int Cout) {
    if (taps < 1 || Cred < 8 || Cred % 8 != 0 || Cout < 1) return 0;
    return (size_t)(Cred / 8) * ((Cout + 31) / 32) * taps * S6T_PLANE;
}
extern "C" int mrdis_s6_filter_image(const float* w, int taps, int Cred, int Cout, void* image, size_t image_bytes, void* stream) {
    const size_t need = mrdis_s6_filter_image_bytes(taps, Cred, Cout);
    if (!w || !image || need == 0 || taps > MRDIS_MAX_TAPS) return MRDIS_EINVAL;
    if (image_bytes < need) return MRDIS_EWORKSPACE;
    if (need >= 0x7fffffffULL || (((uintptr_t)image) & 15) != 0) return MRDIS_EUNSUPPORTED;
    const long long rows = (long long)(need / S6T_ROW);
    const long long nb = (rows + 255) / 256;
    MRDIS_LAUNCH(s6_filter_image_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, (hipStream_t)stream, w, taps, Cred, Cout, (unsigned char*)image);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// instantiations: (KC, WP, WC, XR, WR) -- input pieces per thread <= XR, filter pieces per thread <= WR
#define S6T_CASES(X) X(8, 1, 2, 3, 12) X(8, 2, 2, 5, 12) X(8, 2, 1, 5, 6) X(8, 1, 1, 3, 6) \
                     X(16, 1, 2, 3, 12) X(16, 2, 2, 5, 12) X(16, 2, 1, 5, 6) X(16, 1, 1, 3, 6) X(16, 2, 1, 3, 3) X(16, 1, 2, 6, 12) \
                     X(32, 1, 2, 6, 12) X(32, 2, 2, 6, 12) X(32, 2, 1, 6, 6) X(32, 1, 1, 6, 6)

// Eligibility: fp32 views, a filter image from mrdis_s6_filter_image, reduction axis a multiple of 8, couts a multiple of 4 (16-byte stores), 16-byte
// aligned views; the tile must fit the register-staged pieces and the LDS.  MRDIS_EUNSUPPORTED = not eligible (the caller falls back to the fp32 kernel).
int mrdis_run_s6conv(TapConvParams p, const void* image, int taps_img, int dh_max, int dw_max, hipStream_t s, S6ConvLaunch* defer) {
    if (defer) defer->set = false;
    const long long s6 = mrdis_opt(MRDIS_OPT_SPLIT6);
    if (!(s6 == 1 || s6 == 10) || p.dtype != MRDIS_DT_F32 || !p.in || !image || taps_img < 1 || !p.out) return MRDIS_EUNSUPPORTED;
    if (p.is != 1 && p.is != 2) return MRDIS_EUNSUPPORTED;
    if (p.Cin % 8 != 0 || p.Cout % 4 != 0 || p.Cout < 16) return MRDIS_EUNSUPPORTED;
    if (p.ldin % 4 != 0 || p.ldout % 4 != 0 || (((uintptr_t)p.in | (uintptr_t)image | (uintptr_t)p.out) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (p.bias && (((uintptr_t)p.bias) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)p.N * p.Hin * p.Win * p.ldin >= 0x7fffffffLL * 2LL) return MRDIS_EUNSUPPORTED;
    if (mrdis_s6_filter_image_bytes(taps_img, p.Cin, p.Cout) >= 0x7fffffffULL) return MRDIS_EUNSUPPORTED;
    for (int t = 0; t < p.ntaps; ++t) if (p.widx[t] < 0 || p.widx[t] >= taps_img) return MRDIS_EINVAL;
    p.w_bf16 = image;
    const int ncu = s6t_ncu();
    struct Cfg { int kc, wp, wc, xr, wr; };
    static const Cfg menu[] = {
#define S6T_MENU(KC_, A_, B_, XR_, WR_) {KC_, A_, B_, XR_, WR_},
        S6T_CASES(S6T_MENU)
#undef S6T_MENU
    };
    int TinWp = 0;
    auto geom = [&](int bm) {
        const TileChoice tc = choose_tile(p.N, p.A, p.B, bm);
        p.NB = tc.NB; p.TH = tc.TH; p.TW = tc.TW;
        p.TinH = (p.TH - 1) * p.is + (dh_max - p.dh_min) + 1;
        p.TinW = (p.TW - 1) * p.is + (dw_max - p.dw_min) + 1;
        p.tilesA = mrdis_cdiv(p.A, p.TH); p.tilesB = mrdis_cdiv(p.B, p.TW); p.tilesN = mrdis_cdiv(p.N, p.NB);
        // row pitch of the pixel image: the tile rows inside a 32-lane half (32 / TW of them) must start 0 (TW = 16) or 8 (TW = 8) slots apart mod 16
        TinWp = p.TinW + (p.is == 2 ? (p.TinW & 1) : 0);            // (stride 2: both parity halves of a row have (TinW + 1) / 2 columns)
        if (p.TW == 16 || p.TW == 8) {
            const int want = p.TW == 16 ? 0 : 8;
            while ((p.is * TinWp) % 16 != want) ++TinWp;
        }
    };
    // the configuration, by measured preference (tools/s6conv_abl.py, B = 32 encoder layers): 16-channel chunks before 32 before 8; few taps (the parity
    // classes of a stride-2 data gradient): 64 positions x 32 couts per wave before the square tile before the small ones; many taps (a 4x4 / 3x3 forward):
    // the square tile first (every filter row is read once per 64 positions).  debug_mode = 100 kc + 10 wp + wc forces one.
    const long long force = mrdis_opt(MRDIS_OPT_MODE);
    const Cfg* best = nullptr; size_t best_lds = 0; long long best_score = -1;
    for (const Cfg& c : menu) {
        if (force > 0 && force != 100LL * c.kc + 10 * c.wp + c.wc) continue;
        if (p.Cin % c.kc != 0) continue;
        if (c.wc == 2 && p.Cout <= 32) continue;
        geom(128 * c.wp);
        const int qx = c.kc / 8;
        if (p.TinH >= 256 || p.TinW >= 256 || p.ntaps * qx > MRDIS_MAX_TAPS * 4) continue;
        if ((long long)p.NB * p.TinH * p.TinW * qx > c.xr * 256LL) continue;
        if ((long long)qx * c.wc * p.ntaps * (S6T_PLANE / 16) > c.wr * 256LL) continue;
        const size_t lds = (size_t)qx * c.wc * p.ntaps * S6T_PLANE + (size_t)s6t_pitch(c.kc) * p.NB * p.TinH * TinWp;
        if (lds > 156 * 1024) continue;
        const int kc_rank = c.kc == 16 ? 3 : (c.kc == 32 ? 2 : 1);
        const int tile = 10 * c.wp + c.wc;
        const int tile_rank = p.ntaps <= 4 ? (tile == 21 ? 4 : tile == 22 ? 3 : tile == 11 ? 2 : 1) : (tile == 22 ? 4 : tile == 11 ? 3 : tile == 12 ? 2 : 1);
        const long long score = 1000LL * kc_rank + 100 * tile_rank - (c.xr + c.wr);
        if (score > best_score) { best_score = score; best = &c; best_lds = lds; }
    }
    if (!best) return MRDIS_EUNSUPPORTED;
    const Cfg c = *best;
    geom(128 * c.wp);
    const size_t lds = best_lds;
    p.coTiles = mrdis_cdiv(p.Cout, 32 * c.wc);
    S6ConvGeom g{};
    const long long tiles = (long long)p.tilesA * p.tilesB * p.tilesN;
    if (tiles > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    g.tiles = (int)tiles; g.nchunks = p.Cin / c.kc; g.TinWp = TinWp; g.groups = (p.Cout + 31) / 32; g.taps_img = taps_img;
#ifdef S6T_STAMPS
    g.stamps = reinterpret_cast<unsigned long long*>((uintptr_t)(mrdis_opt(MRDIS_OPT_BM) > 0 ? mrdis_opt(MRDIS_OPT_BM) : 0));      // stamp buffer (device pointer) and stamps per wave, via options debug_bm / debug_bn
    g.cap_stamps = (int)mrdis_opt(MRDIS_OPT_BN);
#endif
    int per_cu = 1;                                // persistent grid: what the CUs hold at once of THIS instantiation (its registers and its LDS)
#define S6T_OCC(KC_, A_, B_, XR_, WR_) if (c.kc == KC_ && c.wp == A_ && c.wc == B_ && c.xr == XR_ && c.wr == WR_) \
        per_cu = defer ? s6t_occ_of(s6conv_pack_kernel<KC_, A_, B_, XR_, WR_>, lds) : s6t_occ_of(s6conv_kernel<KC_, A_, B_, XR_, WR_>, lds);
    S6T_CASES(S6T_OCC)
#undef S6T_OCC
    long long per_cot = (long long)ncu * per_cu / p.coTiles;
    if (defer) per_cot /= 4;                       // the four parity classes of a stride-2 data gradient share one launch
    { const long long v = mrdis_opt(MRDIS_OPT_WGSPLIT); if (v > 0) per_cot = v; }      // debug_wgsplit: workgroups per cout tile
    if (per_cot < 1) per_cot = 1;
    if (per_cot > tiles) per_cot = tiles;
    g.tile_stride = (int)per_cot;
    const int grid = (int)per_cot * p.coTiles;
    if (defer) { defer->p = p; defer->g = g; defer->KC = c.kc; defer->wp = c.wp; defer->wc = c.wc; defer->xr = c.xr; defer->wr = c.wr; defer->grid = grid; defer->lds = lds; defer->set = true; return MRDIS_OK; }
#define S6T_RUN(KC_, A_, B_, XR_, WR_) if (c.kc == KC_ && c.wp == A_ && c.wc == B_ && c.xr == XR_ && c.wr == WR_) return launch_s6conv_t<KC_, A_, B_, XR_, WR_>(p, g, grid, lds, s);
    S6T_CASES(S6T_RUN)
#undef S6T_RUN
    return MRDIS_EUNSUPPORTED;
}

// the planned classes of a stride-2 data gradient: one launch if they agree on the instantiation, else one each
int mrdis_launch_s6conv_planned(const S6ConvLaunch (&L)[4], hipStream_t s) {
    bool same = true;
    for (int k = 0; k < 4; ++k) same = same && L[k].set && L[k].KC == L[0].KC && L[k].wp == L[0].wp && L[k].wc == L[0].wc && L[k].xr == L[0].xr && L[k].wr == L[0].wr;
    if (same) {
#define S6T_RUNP(KC_, A_, B_, XR_, WR_) if (L[0].KC == KC_ && L[0].wp == A_ && L[0].wc == B_ && L[0].xr == XR_ && L[0].wr == WR_) return launch_s6conv_pack_t<KC_, A_, B_, XR_, WR_>(L, s);
        S6T_CASES(S6T_RUNP)
#undef S6T_RUNP
        return MRDIS_EUNSUPPORTED;
    }
    for (int k = 0; k < 4; ++k) {
        if (!L[k].set) continue;
        int rc = MRDIS_EUNSUPPORTED;
#define S6T_RUN1(KC_, A_, B_, XR_, WR_) if (L[k].KC == KC_ && L[k].wp == A_ && L[k].wc == B_ && L[k].xr == XR_ && L[k].wr == WR_) rc = launch_s6conv_t<KC_, A_, B_, XR_, WR_>(L[k].p, L[k].g, L[k].grid, L[k].lds, s);
        S6T_CASES(S6T_RUN1)
#undef S6T_RUN1
        if (rc) return rc;
    }
    return MRDIS_OK;
}
