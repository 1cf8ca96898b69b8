// mrdis_bf16.hip -- the tap-table convolution (mrdis_conv.hip) on v_mfma_f32_32x32x16_bf16: bf16 MFMA operands, fp32
// accumulation, fp32 bias / epilogue.  BASELINE.json configs[2] ("bf16"), stage 1: activations stay fp32 in HBM
// (MRDIS_DT_F32_BF16M) and are rounded to bf16 (RNE) on their way into LDS; the filter arrives as a bf16 copy of the mixed
// kernel with the reduction axis contiguous.  At 16x the fp32 MFMA rate the big layers become HBM-bound, so the kernel is
// built around keeping loads in flight:
//
//   * PERSISTENT workgroups walk (position tile, channel chunk) items; the global loads of item i+1 (input tile chunk and
//     filter chunk, register-staged) are issued before the MFMAs of item i start, and a filter that fits one chunk
//     (Cin <= KC) is staged once per workgroup instead of once per tile;
//   * LDS images: xs[pixel][KC + 8] and ws[tap][cout][KC + 8] bf16 -- a 16-byte ds_read_b128 per lane is one MFMA operand
//     (8 consecutive reduction indices); the 16-byte row pad makes the 16 lanes of a read group land on distinct banks;
//   * operands swapped as in the fp32 kernels (A = filter, B = pixels): D[cout][position], a lane owns one position and
//     4-cout groups, i.e. 16-byte stores.
//
// Reference op: F.conv2d at src/model.py:2104 and its autograd data gradient.
#include "mrdis_tapconv.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned bw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bw_u32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ bf16x8 cvt8(const float4 a, const float4 b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}

// 8 consecutive channels of one pixel in flight between global memory and LDS: two float4 (fp32 storage, converted when
// they are written to LDS) or one 16-byte bf16x8 (bf16 storage)
template <typename TS> struct Piece8;
template <> struct Piece8<float> {
    float4 a, b;
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ bf16x8 bf() const { return cvt8(a, b); }
    __device__ __forceinline__ void add_to(float* s) const { s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w; }
};
template <> struct Piece8<__bf16> {
    bf16x8 v;
    __device__ __forceinline__ void zero() { for (int k = 0; k < 8; ++k) v[k] = (__bf16)0.f; }
    __device__ __forceinline__ void load(const __bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ bf16x8 bf() const { return v; }
    __device__ __forceinline__ void add_to(float* s) const { for (int k = 0; k < 8; ++k) s[k] += (float)v[k]; }
};
__device__ __forceinline__ void store4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(__bf16* p, const float4 v) {
    bf16x4 r; r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(p) = r;
}

// KC channels per chunk; 4 waves as WAVES_C (cout) x 4/WAVES_C (positions); a wave owns WP x WC blocks of 32 positions x 32 couts;
// TS = storage type of the activations (float: MRDIS_DT_F32_BF16M, __bf16: MRDIS_DT_BF16)
// ABL (timing-only builds, -DBCONV_ABLATIONS; results wrong): 1 no MFMAs, 2 no global loads, 4 no LDS stores, 8 no operand reads, 16 no epilogue stores
template <int KC, int WAVES_C, int WP, int WC, typename TS, int ABL = 0>
__device__ __forceinline__ void bconv_body(const TapConvParams& p, const BConvGeom& g, const int bx) {
    constexpr int BN = 32 * WC * WAVES_C;          // x 32 * WP * WAVES_P positions
    constexpr int PITCH = KC + 8;                 // bf16 elements per LDS row
    constexpr int QX = KC / 8;                    // 16-byte pieces per pixel row
    constexpr int XR = 6, WR = 9;                 // register-staged pieces per thread (input, filter)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* ws = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* xs = ws + p.ntaps * BN * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, e = lane & 31;
    const int wave_c = wave % WAVES_C, wave_p = wave / WAVES_C;          // 4 / WAVES_C position groups
    const int cot = bx % p.coTiles, wg = bx / p.coTiles;
    const int co0 = cot * BN;
    const int tinHW = p.TinH * p.TinW, npix_in = p.NB * tinHW, npos = p.NB * p.TH * p.TW;

    // ---- per-lane constants: the pixel slot and (nb, ty, tx) of each of the lane's WP positions
    int abase[WP], pos_nb[WP], pos_ty[WP], pos_tx[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int m = 32 * (wave_p * WP + i) + e;
        int nb = 0, ty = 0, tx = 0;
        if (m < npos) { nb = m / (p.TH * p.TW); const int rem = m - nb * p.TH * p.TW; ty = rem / p.TW; tx = rem - ty * p.TW; }
        else nb = -1;
        pos_nb[i] = nb; pos_ty[i] = ty; pos_tx[i] = tx;
        abase[i] = (nb < 0) ? 0 : ((nb * p.TinH + ty * p.is) * p.TinW + tx * p.is) * PITCH + 8 * half;
    }
    // ---- staging descriptors (tile-invariant part): input pieces and filter pieces of this thread
    int x_desc[XR];                                // (nb << 16) | (iy << 8) | ix of the piece's pixel, or -1
    const int nx = npix_in * QX;
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 256;
        x_desc[it] = -1;
        if (idx < nx) {
            const int pi = idx / QX;
            const int nb = pi / tinHW; const int rem = pi - nb * tinHW;
            const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
            x_desc[it] = (nb << 16) | (iy << 8) | ix;
        }
    }
    const int nw = p.ntaps * BN * QX;
    int w_off[WR];                                 // element offset of the piece in the bf16 filter at chunk 0, or -1 (host: < 2^31)
#pragma unroll
    for (int it = 0; it < WR; ++it) {
        const int idx = tid + it * 256;
        w_off[it] = -1;
        if (idx < nw) {
            const int row = idx / QX, q = idx - row * QX;
            const int t = row / BN, co = co0 + (row - t * BN);
            if (co < p.Cout) w_off[it] = (p.widx[t] * p.Cout + co) * p.Cin + 8 * q;
        }
    }
    const __bf16* wsrc = reinterpret_cast<const __bf16*>(p.w_bf16);

    Piece8<TS> xr[XR];
    bf16x8 wr[WR];
    const TS* in = reinterpret_cast<const TS*>(p.in);
    TS* outp = reinterpret_cast<TS*>(p.out);
    auto tile_origin = [&](int tile, int& n0, int& a0, int& b0) {
        const int tb = tile % p.tilesB; tile /= p.tilesB;
        const int ta = tile % p.tilesA;
        n0 = (tile / p.tilesA) * p.NB; a0 = ta * p.TH; b0 = tb * p.TW;
    };
    auto load_item = [&](int tile, int chunk, bool want_w) {
        int n0, a0, b0; tile_origin(tile, n0, a0, b0);
        const int h_org = a0 * p.is + p.dh_min, w_org = b0 * p.is + p.dw_min, c0 = chunk * KC;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it].zero();
            if (x_desc[it] >= 0) {
                const int n = n0 + (x_desc[it] >> 16), h = h_org + ((x_desc[it] >> 8) & 255), w_ = w_org + (x_desc[it] & 255);
                if (!(ABL & 2) && n < p.N && (unsigned)h < (unsigned)p.Hin && (unsigned)w_ < (unsigned)p.Win)
                    xr[it].load(in + ((long long)(n * p.Hin + h) * p.Win + w_) * p.ldin + c0 + 8 * ((tid + it * 256) % QX));
            }
        }
        if (want_w) {
#pragma unroll
            for (int it = 0; it < WR; ++it) {
                bf16x8 z; for (int k = 0; k < 8; ++k) z[k] = (__bf16)0.f;
                wr[it] = z;
                if (!(ABL & 2) && w_off[it] >= 0) wr[it] = *reinterpret_cast<const bf16x8*>(wsrc + w_off[it] + c0);
            }
        }
    };
    bool first_store = true;
    auto store_item = [&](bool have_w) {
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + it * 256;
            if (idx < nx && !((ABL & 4) && !first_store)) *reinterpret_cast<bf16x8*>(xs + (idx / QX) * PITCH + 8 * (idx % QX)) = xr[it].bf();
        }
        if (have_w) {
#pragma unroll
            for (int it = 0; it < WR; ++it) {
                const int idx = tid + it * 256;
                if (idx < nw && !((ABL & 4) && !first_store)) *reinterpret_cast<bf16x8*>(ws + (idx / QX) * PITCH + 8 * (idx % QX)) = wr[it];
            }
        }
    };

    f32x16 acc[WC][WP];
    const bool w_resident = g.nchunks == 1;       // the whole reduction fits one chunk: the filter is staged once
    int tile = wg, chunk = 0;
    if (tile >= g.tiles) return;
    load_item(tile, 0, true);
    bool first = true;
    const bool lrelu = (p.epilogue & MRDIS_EPI_LRELU) != 0;
    const bool wide8 = std::is_same<TS, __bf16>::value && (p.epilogue & 0x100) != 0;          // host: Cout % 8 == 0, ldout % 8 == 0, 16-byte aligned view, not debug_nopack
    while (tile < g.tiles) {
        __syncthreads();                           // everyone is done reading the previous item's LDS images
        store_item(first || !w_resident);
        __syncthreads();
        // next item: its loads fly while this one is multiplied
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == g.nchunks) { nchunk = 0; ntile = tile + g.tile_stride; }
        if (ntile < g.tiles) load_item(ntile, nchunk, !w_resident);
        first = false;
        if (chunk == 0) {
#pragma unroll
            for (int j = 0; j < WC; ++j)
#pragma unroll
                for (int i = 0; i < WP; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
        }
        const __bf16* wbase = ws + (32 * (wave_c * WC) + e) * PITCH + 8 * half;
        for (int t = 0; t < p.ntaps; ++t) {
            const int toff = ((p.dh[t] - p.dh_min) * p.TinW + (p.dw[t] - p.dw_min)) * PITCH;
            const __bf16* wt = wbase + t * BN * PITCH;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                bf16x8 bf[WP], af[WC];
#pragma unroll
                for (int i = 0; i < WP; ++i) bf[i] = *reinterpret_cast<const bf16x8*>(xs + abase[i] + ((ABL & 8) ? 0 : toff + 16 * ks));
#pragma unroll
                for (int j = 0; j < WC; ++j) af[j] = *reinterpret_cast<const bf16x8*>(((ABL & 8) ? wbase : wt + 16 * ks) + 32 * j * PITCH);
#pragma unroll
                for (int j = 0; j < WC; ++j)
#pragma unroll
                    for (int i = 0; i < WP; ++i) {
                        if (ABL & 1) acc[j][i][0] += (float)af[j][0] * (float)bf[i][0];
                        else acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j], bf[i], acc[j][i], 0, 0, 0);
                    }
            }
        }
        first_store = false;
        if (chunk == g.nchunks - 1) {
            // ---- epilogue: D[cout][position]; a lane owns one position per block and the couts 8g + 4*half .. +3 (g = 0..3)
            int n0, a0, b0; tile_origin(tile, n0, a0, b0);
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                if (pos_nb[i] < 0) continue;
                const int n = n0 + pos_nb[i], a = a0 + pos_ty[i], b = b0 + pos_tx[i];
                if (n >= p.N || a >= p.A || b >= p.B) continue;
                TS* dst = outp + ((long long)(n * p.Hout + a * p.os + p.oh0) * p.Wout + b * p.os + p.ow0) * p.ldout;
                if (wide8) {
                    // bf16 output: the two half-waves of a position swap one 4-cout piece (v_permlane32_swap), every lane then holds 8 consecutive couts -- one
                    // 16-byte store instead of two of 8 (8-byte pieces reach L2 as 16 bytes per line and instruction and halve the write rate:
                    // tools/micro/store_pattern.hip).  Both lanes of a pair own the same position, so they take the same branches above.
#pragma unroll
                    for (int j = 0; j < WC; ++j) {
                        bw_u32x2 pk[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int co = co0 + 32 * (wave_c * WC + j) + 8 * q + 4 * half;
                            float4 v = make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
                            if (p.bias && co < p.Cout) { const float4 bb = *reinterpret_cast<const float4*>(p.bias + co); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                            if (lrelu) { v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y; v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w; }
                            bf16x4 r; r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
                            pk[q] = __builtin_bit_cast(bw_u32x2, r);
                        }
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[q][0], pk[q + 1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[q][1], pk[q + 1][1], false, false);
                            const int co = co0 + 32 * (wave_c * WC + j) + 8 * (q + half);          // this lane's eight consecutive couts
                            if (co < p.Cout && !(ABL & 16)) *reinterpret_cast<bw_u32x4*>(reinterpret_cast<__bf16*>(dst) + co) = bw_u32x4{s0[0], s1[0], s0[1], s1[1]};
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int j = 0; j < WC; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = co0 + 32 * (wave_c * WC + j) + 8 * q + 4 * half;
                        if (co >= p.Cout) continue;
                        float4 v = make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
                        if (p.bias) { const float4 bb = *reinterpret_cast<const float4*>(p.bias + co); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
                        if (lrelu) { v.x = v.x > 0.f ? v.x : 0.2f * v.x; v.y = v.y > 0.f ? v.y : 0.2f * v.y; v.z = v.z > 0.f ? v.z : 0.2f * v.z; v.w = v.w > 0.f ? v.w : 0.2f * v.w; }
                        if (!(ABL & 16) || v.x == 1.2345f) store4(dst + co, v);
                    }
            }
        }
        tile = ntile; chunk = nchunk;
    }
}

template <int KC, int WAVES_C, int WP, int WC, typename TS, int ABL = 0>
__global__ __launch_bounds__(256, 2) void bconv_kernel(const TapConvParams p, const BConvGeom g) {
    bconv_body<KC, WAVES_C, WP, WC, TS, ABL>(p, g, (int)blockIdx.x);
}
// the four output-parity classes of a stride-2 data gradient in one launch (blockIdx.y = class), as tapconv_pack_kernel
struct BConvPack { TapConvParams c[4]; BConvGeom g[4]; int grid[4]; };
template <int KC, int WAVES_C, int WP, int WC, typename TS>
__global__ __launch_bounds__(256, 2) void bconv_pack_kernel(const BConvPack pk) {
    const int y = blockIdx.y;
    if ((int)blockIdx.x >= pk.grid[y]) return;
    bconv_body<KC, WAVES_C, WP, WC, TS, 0>(pk.c[y], pk.g[y], (int)blockIdx.x);
}
template <int KC, int WAVES_C, int WP, int WC, typename TS>
static int launch_bconv_pack_t(const BConvLaunch (&L)[4], hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)bconv_pack_kernel<KC, WAVES_C, WP, WC, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    BConvPack pk;
    int gx = 0; size_t lds = 0;
    for (int k = 0; k < 4; ++k) { pk.c[k] = L[k].p; pk.g[k] = L[k].g; pk.grid[k] = L[k].grid; if (L[k].grid > gx) gx = L[k].grid; if (L[k].lds > lds) lds = L[k].lds; }
    MRDIS_LAUNCH((bconv_pack_kernel<KC, WAVES_C, WP, WC, TS>), dim3(gx, 4), dim3(256), lds, s, pk);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

template <int KC, int WAVES_C, int WP, int WC, typename TS>
static int launch_bconv_t(const TapConvParams& p, const BConvGeom& g, int grid, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)bconv_kernel<KC, WAVES_C, WP, WC, TS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
#ifdef BCONV_ABLATIONS
    if (KC == 32 && WP == 2 && WC == 2) {
        const int abl = (int)mrdis_opt(MRDIS_OPT_MODE);
#define BA(a) if (abl == a) { hipFuncSetAttribute((const void*)bconv_kernel<KC, WAVES_C, WP, WC, TS, a>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
        MRDIS_LAUNCH((bconv_kernel<KC, WAVES_C, WP, WC, TS, a>), dim3(grid), dim3(256), lds, s, p, g); MRDIS_CHECK_LAUNCH(); return MRDIS_OK; }
        BA(1) BA(2) BA(4) BA(8) BA(16) BA(6) BA(14) BA(15) BA(30)
#undef BA
    }
#endif
    MRDIS_LAUNCH((bconv_kernel<KC, WAVES_C, WP, WC, TS>), dim3(grid), dim3(256), lds, s, p, g);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
template <int KC, int WAVES_C, int WP, int WC>
static int launch_bconv(const TapConvParams& p, const BConvGeom& g, int grid, size_t lds, hipStream_t s) {
    return p.dtype == MRDIS_DT_BF16 ? launch_bconv_t<KC, WAVES_C, WP, WC, __bf16>(p, g, grid, lds, s)
                                    : launch_bconv_t<KC, WAVES_C, WP, WC, float>(p, g, grid, lds, s);
}

static int bconv_ncu() {
    static int ncu = 0;
    if (!ncu) {
        hipDeviceProp_t prop; int dev = 0; (void)hipGetDevice(&dev);
        ncu = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

int mrdis_run_bconv3(const TapConvParams& t, hipStream_t s);      // mrdis_bf16p.hip
int mrdis_run_bconv4(const TapConvParams& t, hipStream_t s);      // mrdis_bf16q.hip

// Eligibility: reduction axis a multiple of 16, 16-byte aligned views, cout a multiple of 4 (16-byte stores).
int mrdis_run_bconv(TapConvParams p, int dh_max, int dw_max, hipStream_t s, BConvLaunch* defer) {
    if (defer) defer->set = false;
    if (!p.w_bf16 || (p.dtype != MRDIS_DT_F32_BF16M && p.dtype != MRDIS_DT_BF16)) return MRDIS_EUNSUPPORTED;
    const bool st_bf16 = p.dtype == MRDIS_DT_BF16;
    if (st_bf16 && mrdis_opt(MRDIS_OPT_WINO_PIPE)) {               // 3x3 s1 layers on bf16 activations: the pipelined kernel (mrdis_bf16p.hip)
        const int rc4 = mrdis_run_bconv4(p, s);                    // LDS-DMA form where the launch fills the chip
        if (rc4 != MRDIS_EUNSUPPORTED) return rc4;
        const int rc3 = mrdis_run_bconv3(p, s);
        if (rc3 != MRDIS_EUNSUPPORTED) return rc3;
    }
    if (p.Cin % 16 != 0 || p.Cout % 4 != 0 || p.Cout < 16 || (long long)MRDIS_MAX_TAPS * p.Cin * p.Cout >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    // 16-byte input pieces (8 bf16 or 2 x 4 fp32), 4-channel output stores
    if (p.ldin % (st_bf16 ? 8 : 4) != 0 || p.ldout % 4 != 0 || (((uintptr_t)p.in | (uintptr_t)p.w_bf16) & 15) != 0 ||
        ((uintptr_t)p.out & (st_bf16 ? 7 : 15)) != 0) return MRDIS_EUNSUPPORTED;
    if (p.bias && (((uintptr_t)p.bias) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (st_bf16 && p.Cout % 8 == 0 && p.ldout % 8 == 0 && (((uintptr_t)p.out) & 15) == 0 && !mrdis_opt(MRDIS_OPT_NOPACK)) p.epilogue |= 0x100;      // 16-byte output stores
    // tile shape by cout width: (WAVES_C, WP, WC) -> BM x BN
    struct Cfg { int waves_c, wp, wc; };
    Cfg c;
    // measured (tools/layer_bench.py --dtype bf16): 64-cout tiles with 32-channel chunks beat 128-cout tiles (which only fit
    // 16-channel chunks in LDS) on every wide layer: sp4.gamma+beta 153 -> 123 us, ana.up_3 106 -> 83 us
    if (p.Cout > 32) c = {1, 2, 2};               // 256 positions x 64 couts
    else c = {1, 2, 1};                           // 256 positions x 32 couts
    // small maps (the 8x8 / 16x16 levels, B = 32: 2,048 / 8,192 positions): the layer is a latency chain per workgroup, not throughput --
    // fewer than two workgroups per CU with the big tile means idle CUs AND a long chain, so the tile shrinks to 32 couts, then to 128
    // positions (tools/tiny_probe.py: 128 -> 256 at 8x8 29 -> 16 us forward, 46 -> 22 us data gradient; 256 -> 256 at 16x16 38 -> 27 us)
    {
        const long long md = mrdis_opt(MRDIS_OPT_MODE);
        auto blocks = [&](const Cfg& k) {
            const long long bm = 32LL * k.wp * (4 / k.waves_c), bn = 32LL * k.wc * k.waves_c;
            return (((long long)p.N * p.A * p.B + bm - 1) / bm) * ((p.Cout + bn - 1) / bn);
        };
        if (md != 0 && (long long)p.N * p.A * p.B <= 16384) {      // (64 -> 128 stride 2 onto 32x32 re-stages its big halo tile per cout tile: 44 -> 53 us)
            if (c.wc == 2 && blocks(c) < 2LL * bconv_ncu()) c.wc = 1;
            if (c.wp == 2 && blocks(c) < 2LL * bconv_ncu()) c.wp = 1;
        }
    }
    int BM = 32 * c.wp * (4 / c.waves_c);          // positions per workgroup
    const int BN = 32 * c.wc * c.waves_c;
    auto geom = [&](int bm) {
        const TileChoice tc = choose_tile(p.N, p.A, p.B, bm);
        p.NB = tc.NB; p.TH = tc.TH; p.TW = tc.TW;
        p.TinH = (p.TH - 1) * p.is + (dh_max - p.dh_min) + 1;
        p.TinW = (p.TW - 1) * p.is + (dw_max - p.dw_min) + 1;
        p.tilesA = mrdis_cdiv(p.A, p.TH); p.tilesB = mrdis_cdiv(p.B, p.TW); p.tilesN = mrdis_cdiv(p.N, p.NB);
    };
    geom(BM);
    int KC = (p.Cin % 32 == 0) ? 32 : 16;
    auto npix = [&]() { return (long long)p.NB * p.TinH * p.TinW; };
    auto lds_bytes = [&](int kc) { return 2 * (size_t)(kc + 8) * ((size_t)p.ntaps * BN + (size_t)npix()); };
    // register-staged pieces per thread (6 input, 9 filter) and LDS: two workgroups per CU up to 80 KB each, one up to 120 KB
    auto fits = [&](int kc, size_t cap) { return p.TinH < 256 && p.TinW < 256 && npix() * (kc / 8) <= 6 * 256 && (long long)p.ntaps * BN * (kc / 8) <= 9 * 256 && lds_bytes(kc) <= cap; };
    auto pick = [&](size_t cap) {
        KC = (p.Cin % 32 == 0) ? 32 : 16;
        if (!fits(KC, cap) && KC == 32) KC = 16;
        return fits(KC, cap);
    };
    bool ok = pick(80 * 1024);
    if (!ok && c.wp == 2) {                                  // stride-2 halos, many-image tiles of tiny maps: 128-position tiles
        c.wp = 1; BM = 128; geom(BM);
        ok = pick(80 * 1024) || pick(120 * 1024);
    }
    if (!ok) ok = pick(120 * 1024);
    if (!ok) return MRDIS_EUNSUPPORTED;
    if ((long long)p.N * p.Hin * p.Win * p.ldin >= 0x7fffffffLL * 2LL) return MRDIS_EUNSUPPORTED;
    p.coTiles = mrdis_cdiv(p.Cout, BN);
    BConvGeom g;
    const long long tiles = (long long)p.tilesA * p.tilesB * p.tilesN;
    if (tiles > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    g.tiles = (int)tiles; g.nchunks = p.Cin / KC;
    const size_t lds = lds_bytes(KC);
    int per_cu = (int)((150 * 1024) / (lds + 1024)); if (per_cu > 2) per_cu = 2; if (per_cu < 1) per_cu = 1;
    long long per_cot = (long long)bconv_ncu() * per_cu / p.coTiles; if (per_cot < 1) per_cot = 1;
    if (per_cot > tiles) per_cot = tiles;
    g.tile_stride = (int)per_cot;
    const int grid = (int)per_cot * p.coTiles;
    if (defer) { defer->p = p; defer->g = g; defer->KC = KC; defer->waves_c = c.waves_c; defer->wp = c.wp; defer->wc = c.wc; defer->grid = grid; defer->lds = lds; defer->set = true; return MRDIS_OK; }
#define BC_CASE(kc, a, b_, d) if (KC == kc && c.waves_c == a && c.wp == b_ && c.wc == d) return launch_bconv<kc, a, b_, d>(p, g, grid, lds, s)
    BC_CASE(32, 1, 2, 2); BC_CASE(16, 1, 2, 2); BC_CASE(32, 1, 1, 2); BC_CASE(16, 1, 1, 2);
    BC_CASE(32, 1, 2, 1); BC_CASE(16, 1, 2, 1); BC_CASE(32, 1, 1, 1); BC_CASE(16, 1, 1, 1);
#undef BC_CASE
    return MRDIS_EUNSUPPORTED;
}

// ------------------------------------------------------------------ fp32 -> bf16 (round to nearest even), e.g. the mixed filters
__global__ void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        bf16x4 r; r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(dst + i) = r;
    } else {
        for (long long k = i; k < n; ++k) dst[k] = (__bf16)src[k];
    }
}
extern "C" int mrdis_cast_bf16(const float* src, void* dst, long long n, void* stream) {
    if (!src || !dst || n < 1 || (((uintptr_t)src & 15) != 0) || (((uintptr_t)dst & 7) != 0)) return MRDIS_EINVAL;
    MRDIS_LAUNCH(cast_bf16_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<__bf16*>(dst), n);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// =====================================================================================================================
// Weight gradient on bf16 MFMA: dw[t][ci][co] = sum over (n, a, b) of x[n, a + dh_t, b + dw_t, ci] * dy[n, a, b, co]
// (stride 1; autograd convolution_backward, weight part, of F.conv2d at src/model.py:2104).
//
// The reduction axis is the POSITION, and both operands are stored channel-contiguous (NHWC), so the MFMA operands are
// gathered with gfx950's transposing LDS read: the tiles sit in LDS as [pixel][32 channels] bf16 images (64-byte rows: the
// four rows of one ds_read_b64_tr_b16 block cover all 64 banks once) and a lane receives 4 consecutive positions of one
// channel per read -- two reads per 32x32x16 operand.  A workgroup (512 threads, 8 waves) owns CIW x COW channels and a
// grid-stride share of the 128-position tiles; a wave keeps the 32 x 32 block of ALL taps (9 x 16 accumulators), so one dy
// operand feeds 9 MFMAs.  Waves that share a channel block split the 8 position steps of a tile among themselves.
// Global loads of the next tile (register-staged, fp32 -> bf16 on the way into LDS) fly while the current one is
// multiplied; the bias gradient (column sums of dy) is accumulated in fp32 from the staging registers.  Every wave writes
// its partial; the waves of a channel block add theirs through LDS in a fixed order and the workgroup writes one slab
// [split][T][Ci][Co]; a fixed-order reduction kernel sums the slabs (bit-reproducible).
struct BWgradParams {
    const void* x; const void* dy; float* slab; float* bias_slab;      // x, dy: fp32 or bf16 views (kernel template)
    int N, H, W, Ci, Co, ldx, lddy;
    int ntaps, dh[9], dw[9];
    int NB, TH, TW, lgTH, lgTW, TinH, TinW;         // TH, TW powers of two, NB * TH * TW = 128
    int tilesA, tilesB, tilesN, tiles;
    int nCiB, nCoB, splits;
    int dh_min, dw_min;
    int rsx, isx;                                   // x row / image strides in elements (W * ldx, H * W * ldx; a stride-2 parity class: twice that)
    int tstride, tbase;                             // slab [split][tstride taps][Ci][Co], this launch's taps start at tbase
};
struct BWgradPack { BWgradParams c[4]; };           // the four input-parity classes of a stride-2 layer (blockIdx.y)

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int WCI, int WCO, typename TS, bool HALF>      // HALF: Ci = 16, the upper half of the 32-channel image stays zero
__device__ __forceinline__ void bwgrad_body(const BWgradParams& p) {
    constexpr int KS = 8 / (WCI * WCO);            // waves sharing a channel block split the position steps
    constexpr int CIW = 32 * WCI, COW = 32 * WCO;
    constexpr int XQ = CIW / 8, YQ = COW / 8;       // 16-byte pieces per pixel
    constexpr int XR = (WCI == 2) ? 4 : 2, YR = (WCO == 2) ? 2 : 1;
    constexpr int TP = 128;                         // positions per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* dys = reinterpret_cast<__bf16*>(smem_raw);                  // [WCO][128][32]
    __bf16* xs = dys + WCO * TP * 32;                                    // [WCI][npix][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wci = wave % WCI, wco = (wave / WCI) % WCO, ks = wave / (WCI * WCO);
    int bid = blockIdx.x;
    const int split = bid % p.splits; bid /= p.splits;
    const int cob = bid % p.nCoB, cib = bid / p.nCoB;
    const int ci0 = cib * CIW, co0 = cob * COW;
    const int tinHW = p.TinH * p.TinW, npix = p.NB * tinHW;

    // lane constants of the transposing reads: 16-lane group g, row q, 4-column piece pc
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int chan = 16 * (g & 1) + 4 * pc, kbase = 8 * (g >> 1) + q;

    int x_desc[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 512;
        x_desc[it] = -1;
        if (idx < npix * XQ) {
            const int pi = idx / XQ;
            const int nb = pi / tinHW; const int rem = pi - nb * tinHW;
            const int iy = rem / p.TinW, ix = rem - iy * p.TinW;
            x_desc[it] = (nb << 16) | (iy << 8) | ix;
        }
    }
    Piece8<TS> xr[XR], yr[YR];
    const TS* xin = reinterpret_cast<const TS*>(p.x);
    const TS* dyin = reinterpret_cast<const TS*>(p.dy);
    float bsum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] = 0.f;
    auto tile_origin = [&](int tile, int& n0, int& a0, int& b0) {
        const int tb = tile % p.tilesB; tile /= p.tilesB;
        const int ta = tile % p.tilesA;
        n0 = (tile / p.tilesA) * p.NB; a0 = ta * p.TH; b0 = tb * p.TW;
    };
    auto load_tile = [&](int tile) {
        int n0, a0, b0; tile_origin(tile, n0, a0, b0);
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it].zero();
            if (x_desc[it] >= 0) {
                const int n = n0 + (x_desc[it] >> 16), h = a0 + p.dh_min + ((x_desc[it] >> 8) & 255), w_ = b0 + p.dw_min + (x_desc[it] & 255);
                const int cq = ci0 + 8 * ((tid + it * 512) % XQ);
                if (n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W && (!HALF || cq < p.Ci))
                    xr[it].load(xin + (long long)n * p.isx + (long long)h * p.rsx + w_ * p.ldx + cq);
            }
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int idx = tid + it * 512;                 // < 128 * YQ by construction
            const int m = idx / YQ, c = co0 + 8 * (idx % YQ);
            const int tx = m & (p.TW - 1), ty = (m >> p.lgTW) & (p.TH - 1), nb = m >> (p.lgTW + p.lgTH);
            const int n = n0 + nb, a = a0 + ty, b = b0 + tx;
            yr[it].zero();
            if (n < p.N && a < p.H && b < p.W && c < p.Co) yr[it].load(dyin + ((long long)(n * p.H + a) * p.W + b) * p.lddy + c);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int idx = tid + it * 512;
            if (idx < npix * XQ) {
                const int pi = idx / XQ, qq = idx % XQ;
                *reinterpret_cast<bf16x8*>(xs + ((qq >> 2) * npix + pi) * 32 + 8 * (qq & 3)) = xr[it].bf();
            }
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int idx = tid + it * 512;
            const int m = idx / YQ, qq = idx % YQ;
            *reinterpret_cast<bf16x8*>(dys + ((qq >> 2) * TP + m) * 32 + 8 * (qq & 3)) = yr[it].bf();
            yr[it].add_to(bsum);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const __bf16* xw = xs + wci * npix * 32 + chan;
    const __bf16* yw = dys + wco * TP * 32 + chan;
    int tile = split;
    if (tile < p.tiles) load_tile(tile);
    while (tile < p.tiles) {
        __syncthreads();
        store_tile();
        __syncthreads();
        const int ntile = tile + p.splits;
        if (ntile < p.tiles) load_tile(ntile);
        for (int kk = ks; kk < TP / 16; kk += KS) {
            // the lane's two rows (positions) of this step: m0 = 16 kk + kbase and m0 + 4
            const int m0 = 16 * kk + kbase, m1 = m0 + 4;
            const int tx0 = m0 & (p.TW - 1), ty0 = (m0 >> p.lgTW) & (p.TH - 1), nb0 = m0 >> (p.lgTW + p.lgTH);
            const int tx1 = m1 & (p.TW - 1), ty1 = (m1 >> p.lgTW) & (p.TH - 1), nb1 = m1 >> (p.lgTW + p.lgTH);
            const int px0 = (nb0 * p.TinH + ty0) * p.TinW + tx0, px1 = (nb1 * p.TinH + ty1) * p.TinW + tx1;
            union { bf16x8 v; s16x4 h[2]; } bq;
            bq.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yw + m0 * 32));
            bq.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yw + m1 * 32));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t < p.ntaps) {
                    const int off = (p.dh[t] - p.dh_min) * p.TinW + (p.dw[t] - p.dw_min);
                    union { bf16x8 v; s16x4 h[2]; } aq;
                    aq.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(xw + (px0 + off) * 32));
                    aq.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(xw + (px1 + off) * 32));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq.v, bq.v, acc[t], 0, 0, 0);
                }
            }
        }
        tile = ntile;
    }
    // ---- the KS waves of a channel block add their partials through LDS (tap by tap, fixed order), then the ks = 0 wave
    // writes the block: D[ci row][co col] -> slab [split][T][Ci][Co]
    const int half = lane >> 5, e = lane & 31;
    const int co = co0 + 32 * wco + e;
    float* red_acc = reinterpret_cast<float*>(smem_raw);            // [8 waves][16][64]
    // (the tap loop is unrolled so that acc[t] is a fixed register block: with a run-time t every one of the 16 values went through a nine-way
    //  branch chain, twice per tap -- 43 us of a 62-127 us launch, on every layer)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t >= p.ntaps) continue;                    // (uniform; not `break`: the loop must unroll completely)
        __syncthreads();
        if (ks > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red_acc[(wave * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (ks == 0 && co < p.Co) {
            float* dst = p.slab + (((long long)split * p.tstride + p.tbase + t) * p.Ci + ci0 + 32 * wci) * p.Co + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r];
#pragma unroll
                for (int k = 1; k < KS; ++k) v += red_acc[((wave + k * WCI * WCO) * 16 + r) * 64 + lane];
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (!HALF || ci0 + 32 * wci + row < p.Ci) dst[(long long)row * p.Co] = v;
            }
        }
    }
    // ---- bias gradient: column sums of dy from the staging registers (fp32), workgroups of the first ci block only
    if (p.bias_slab && cib == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);            // [512][8]
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bsum[k];
        __syncthreads();
        if (tid < COW) {
            const int qq = tid / 8, k = tid % 8;                    // channel tid = 8 qq + k; threads with tid % YQ == qq hold it
            float sum = 0.f;
            for (int th = qq; th < 512; th += YQ) sum += red[th * 8 + k];
            if (co0 + tid < p.Co) p.bias_slab[(long long)split * p.Co + co0 + tid] = sum;
        }
    }
}

template <int WCI, int WCO, typename TS, bool HALF = false>
__global__ __launch_bounds__(512) void bwgrad_kernel(const BWgradParams p) { bwgrad_body<WCI, WCO, TS, HALF>(p); }

// stride-2 layers (4x4 / 3x3, pad 1: the encoders) on bf16 activations.  Input row 2a + r - pad of tap row r lies in the parity class
// (r - pad) & 1 of the input rows, at row a + ((r - pad) >> 1) of that class: within one class (rows AND columns) the layer is a
// stride-1 "same" correlation of the class view x[:, p::2, q::2] (pixel stride 2 ldx, row stride 2 W ldx) with dy over a 1x1 .. 2x2 window,
// i.e. what bwgrad_body computes.  ONE launch, blockIdx.y = class; the classes share the tile shape, the split count and the slab
// [split][all kh * kw taps, class-major][Ci][Co]; the reduction launch puts tap k of the slab at its place `map[k]` of dw.
template <int WCI, int WCO, bool HALF>
__global__ __launch_bounds__(512) void bwgrad_pack_kernel(const BWgradPack pk) { bwgrad_body<WCI, WCO, __bf16, HALF>(pk.c[blockIdx.y]); }

// ---------------------------------------------------------------------------------------------------------------------
// bwgrad2_kernel: bwgrad_kernel for bf16 activations with the pipeline of bconv3_kernel / wino_wgrad2_kernel.  bwgrad_kernel stages a
// 128-position tile as barrier | registers -> LDS | barrier | next tile's loads | MFMAs; its matrix pipe is 14-16 % busy and the
// big layers run at 2-4x their HBM floor.  Here the two LDS images are double-buffered with ONE barrier per tile: while tile i is
// multiplied, tile i+1 goes from registers to LDS and the loads of tile i+2 are issued, one per MFMA; loads are buffer loads with
// out-of-range offsets for padding / ragged tiles (no branches: counted vmcnt); the lane constants of the transposing reads and the
// staging offsets are tile-invariant.  Same products, same order per workgroup: results identical to bwgrad_kernel's.
template <int V_> struct BwIC { static constexpr int value = V_; };
constexpr unsigned BW_OOB = 0xfffffff0u;
__device__ __forceinline__ int bw_opaque(int idx) { asm volatile("" : "+v"(idx)); return idx; }

#ifndef BW2_AHEAD
#define BW2_AHEAD 1
#endif
template <int WCI, int WCO>
__global__ __launch_bounds__(512, 1) void bwgrad2_kernel(const BWgradParams p, const unsigned x_bytes, const unsigned dy_bytes) {
    constexpr int KS = 8 / (WCI * WCO), NSTEP = 8 / KS;
    constexpr int CIW = 32 * WCI, COW = 32 * WCO;
    constexpr int XQ = CIW / 8, YQ = COW / 8;
    constexpr int XR = (WCI == 2) ? 4 : 2, YR = (WCO == 2) ? 2 : 1, NL = XR + YR, NM = NSTEP * 9;
    constexpr int TP = 128, DYS = WCO * TP * 32;
    static_assert(2 * NL <= NM, "one staging operation per MFMA slot");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem_raw);      // [2][DYS] dy images, then [2][XS] x images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wci = wave % WCI, wco = (wave / WCI) % WCO, ks = wave / (WCI * WCO);
    int bid = blockIdx.x;
    const int split = bid % p.splits; bid /= p.splits;
    const int cob = bid % p.nCoB, cib = bid / p.nCoB;
    const int ci0 = cib * CIW, co0 = cob * COW;
    const int tinHW = p.TinH * p.TinW, npix = p.NB * tinHW;
    const int XS = WCI * npix * 32;

    // lane constants of the transposing reads: 16-lane group g, row q, 4-column piece pc; the lane's two positions of each step
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int chan = 16 * (g & 1) + 4 * pc, kbase = 8 * (g >> 1) + q;
    int xa0[NSTEP], xa1[NSTEP], ya0[NSTEP];            // element offsets in the x image (tap 0 corner) and the dy image
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) {
        const int m0 = 16 * (ks + s_ * KS) + kbase, m1 = m0 + 4;
        const int tx0 = m0 & (p.TW - 1), ty0 = (m0 >> p.lgTW) & (p.TH - 1), nb0 = m0 >> (p.lgTW + p.lgTH);
        const int tx1 = m1 & (p.TW - 1), ty1 = (m1 >> p.lgTW) & (p.TH - 1), nb1 = m1 >> (p.lgTW + p.lgTH);
        xa0[s_] = (wci * npix + (nb0 * p.TinH + ty0) * p.TinW + tx0) * 32 + chan;
        xa1[s_] = (wci * npix + (nb1 * p.TinH + ty1) * p.TinW + tx1) * 32 + chan;
        ya0[s_] = (wco * TP + m0) * 32 + chan;          // m1 = m0 + 4: + 128 elements
    }
    // the host guarantees the canonical 3x3 tap order (tap t = row t / 3, column t % 3 of the window: mrdis_run_bwgrad): a tap's offset in the x
    // image is (t / 3) * rowp + (t % 3) * 32 elements -- three row bases per step and lane, the column part is the read's immediate offset
    // (one address add per tap and operand half, 72 per tile, became 24; and no `t < ntaps` branch around every MFMA)
    const int rowp = p.TinW * 32;

    // staging roles (tile-invariant)
    int x_lds[XR], x_yx[XR]; unsigned x_rel[XR];      // x_yx = (nb << 16) | (iy << 8) | ix, or -1
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * 512, pi = idx / XQ, qq = idx - pi * XQ;
        const int nb = pi / tinHW, rem = pi - nb * tinHW, iy = rem / p.TinW, ix = rem - iy * p.TinW;
        const bool on = idx < npix * XQ && ci0 + 8 * qq < p.Ci;
        x_yx[it] = on ? ((nb << 16) | (iy << 8) | ix) : -1;
        x_lds[it] = ((qq >> 2) * npix + pi) * 32 + 8 * (qq & 3);
        x_rel[it] = 2u * (unsigned)(((nb * p.H + iy) * p.W + ix) * p.ldx + ci0 + 8 * qq);
    }
    int y_lds[YR], y_pos[YR]; unsigned y_rel[YR];     // y_pos = (nb << 16) | (ty << 8) | tx, or -1
#pragma unroll
    for (int it = 0; it < YR; ++it) {
        const int idx = tid + it * 512, m = idx / YQ, qq = idx - m * YQ;      // < 128 * YQ by construction
        const int tx = m & (p.TW - 1), ty = (m >> p.lgTW) & (p.TH - 1), nb = m >> (p.lgTW + p.lgTH);
        y_pos[it] = (co0 + 8 * qq < p.Co) ? ((nb << 16) | (ty << 8) | tx) : -1;
        y_lds[it] = ((qq >> 2) * TP + m) * 32 + 8 * (qq & 3);
        y_rel[it] = 2u * (unsigned)(((nb * p.H + ty) * p.W + tx) * p.lddy + co0 + 8 * qq);
    }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, dy_bytes, 0x00020000);

    bw_u32x4 xr[2][XR], yr[2][YR];
    unsigned xo[XR], yo[YR];
    int ltile = split;                                 // load cursor
    auto next_offsets = [&]() {
        const bool live = ltile < p.tiles;
        int t_ = live ? ltile : 0;
        const int tb = t_ % p.tilesB; t_ /= p.tilesB;
        const int ta = t_ % p.tilesA;
        const int n0 = (t_ / p.tilesA) * p.NB, a0 = ta * p.TH, b0 = tb * p.TW;
        const int h0 = a0 + p.dh_min, w0 = b0 + p.dw_min;
        const unsigned xorg = 2u * (unsigned)(((n0 * p.H + h0) * p.W + w0) * p.ldx);     // may wrap for halo origins: added mod 2^32
        const unsigned yorg = 2u * (unsigned)(((n0 * p.H + a0) * p.W + b0) * p.lddy);
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int n = n0 + (x_yx[it] >> 16), h = h0 + ((x_yx[it] >> 8) & 255), w_ = w0 + (x_yx[it] & 255);
            const bool ok = live && x_yx[it] >= 0 && n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            xo[it] = ok ? xorg + x_rel[it] : BW_OOB;
        }
#pragma unroll
        for (int it = 0; it < YR; ++it) {
            const int n = n0 + (y_pos[it] >> 16), a = a0 + ((y_pos[it] >> 8) & 255), b = b0 + (y_pos[it] & 255);
            const bool ok = live && y_pos[it] >= 0 && n < p.N && a < p.H && b < p.W;
            yo[it] = ok ? yorg + y_rel[it] : BW_OOB;
        }
        ltile += p.splits;
    };
    auto load1 = [&](auto S_, int k) {                  // staging operation k of NL: x pieces first, then dy pieces
        constexpr int S = decltype(S_)::value;
        if (k < XR) xr[S][k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo[k], 0, 0);
        else yr[S][k - XR] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)yo[k - XR], 0, 0);
    };
    float bsum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] = 0.f;
    auto store1 = [&](auto S_, __bf16* dys, __bf16* xs, int k) {
        constexpr int S = decltype(S_)::value;
        if (k < XR) {
            if (tid + k * 512 < npix * XQ) *reinterpret_cast<bw_u32x4*>(xs + x_lds[k]) = xr[S][k];
        } else {
            *reinterpret_cast<bw_u32x4*>(dys + y_lds[k - XR]) = yr[S][k - XR];
            union { bw_u32x4 u; bf16x8 v; } c; c.u = yr[S][k - XR];
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum[j] += (float)c.v[j];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- prologue: tile 0 in LDS buffer 0, tile 1 in register set 1, cursor at tile 2
    next_offsets();
#pragma unroll
    for (int k = 0; k < NL; ++k) load1(BwIC<0>{}, k);
    next_offsets();
#pragma unroll
    for (int k = 0; k < NL; ++k) load1(BwIC<1>{}, k);
#pragma unroll
    for (int k = 0; k < NL; ++k) store1(BwIC<0>{}, lds, lds + 2 * DYS, k);
    __syncthreads();

    const int ntile = (p.tiles - split + p.splits - 1) / p.splits;
    auto iteration = [&](auto P_) {
        constexpr int P = decltype(P_)::value;
        const __bf16* yb = lds + bw_opaque(P * DYS);
        const __bf16* xb = lds + bw_opaque(2 * DYS + P * XS);
        __bf16* dyn = lds + (P ^ 1) * DYS;
        __bf16* xn = lds + 2 * DYS + (P ^ 1) * XS;
        next_offsets();                               // tile i + 2 -> register set P, one load per MFMA slot
        // operands one slot ahead of the MFMA that uses them (two register sets; the sched_barriers keep hipcc from sinking the
        // reads back next to their use: left alone it emits read, read, wait, MFMA and every MFMA eats a full LDS round trip)
        union Op { bf16x8 v; s16x4 h[2]; };
        constexpr int AD = BW2_AHEAD;                 // slots between an A operand's LDS read and the MFMA that uses it
        Op aq[AD + 1], bq[2];
        const __bf16* r0 = xb; const __bf16* r1 = xb;   // row bases of the current (step, tap row): set when a tap row starts
        auto read_a = [&](Op& o, int s_, int t) {
            if (t % 3 == 0) { r0 = xb + xa0[s_] + (t / 3) * rowp; r1 = xb + xa1[s_] + (t / 3) * rowp; }
            o.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(r0 + 32 * (t % 3)));
            o.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(r1 + 32 * (t % 3)));
        };
        auto read_b = [&](Op& o, int s_) {
            o.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yb + ya0[s_]));
            o.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(yb + ya0[s_] + 128));
        };
        read_b(bq[0], 0);
#pragma unroll
        for (int k = 0; k < AD; ++k) read_a(aq[k], k / 9, k % 9);
#pragma unroll
        for (int slot = 0; slot < NM; ++slot) {
            const int s_ = slot / 9, t = slot - 9 * s_;
            if (slot + AD < NM) {
                const int s1 = (slot + AD) / 9, t1 = (slot + AD) - 9 * s1;
                read_a(aq[(slot + AD) % (AD + 1)], s1, t1);
            }
            if (slot + 1 < NM && (slot + 1) % 9 == 0) read_b(bq[((slot + 1) / 9) & 1], (slot + 1) / 9);
            if (slot < NL) load1(BwIC<P>{}, slot);
            if (slot >= NM - NL) store1(BwIC<P ^ 1>{}, dyn, xn, slot - (NM - NL));
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[slot % (AD + 1)].v, bq[s_ & 1].v, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int i = 0; i < ntile; i += 2) {
        iteration(BwIC<0>{});
        if (i + 1 < ntile) iteration(BwIC<1>{});
    }

    // ---- the KS waves of a channel block add their partials through LDS (tap by tap, fixed order), then the ks = 0 wave
    // writes the block: D[ci row][co col] -> slab [split][T][Ci][Co]   (as bwgrad_kernel)
    const int half = lane >> 5, e = lane & 31;
    const int co = co0 + 32 * wco + e;
    float* red_acc = reinterpret_cast<float*>(smem_raw);            // [8 waves][16][64]
#pragma unroll
    for (int t = 0; t < 9; ++t) {                      // unrolled: acc[t] is a fixed register block (see bwgrad_body); the host launches this kernel with 9 taps only
        __syncthreads();                               // (three taps per barrier pair: no faster -- the 37 MB of slab stores per launch are the cost)
        if (ks > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red_acc[(wave * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (ks == 0 && co < p.Co) {
            float* dst = p.slab + (((long long)split * 9 + t) * p.Ci + ci0 + 32 * wci) * p.Co + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r];
#pragma unroll
                for (int k = 1; k < KS; ++k) v += red_acc[((wave + k * WCI * WCO) * 16 + r) * 64 + lane];
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                dst[(long long)row * p.Co] = v;
            }
        }
    }
    if (p.bias_slab && cib == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);            // [512][8]
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bsum[k];
        __syncthreads();
        if (tid < COW) {
            const int qq = tid / 8, k = tid % 8;                    // channel tid = 8 qq + k; threads with tid % YQ == qq hold it
            float sum = 0.f;
            for (int th = qq; th < 512; th += YQ) sum += red[th * 8 + k];
            if (co0 + tid < p.Co) p.bias_slab[(long long)split * p.Co + co0 + tid] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// bwgrad3_kernel (round 5): bwgrad2_kernel with both images staged by LDS-DMA (`buffer_load_dwordx4 ... lds`) into a ring of THREE stages.
// bwgrad2 keeps two tiles in LDS and a third in registers (XR + YR 16-byte loads per thread, each followed by a ds_write_b128); the same diagnosis as for
// bconv3_kernel applies (mrdis_bf16q.hip): the LDS stores and the waits of the MFMA stream on loads, not the matrix pipe, bound it.  Here a tile goes from
// HBM / L2 straight into its stage: no staging registers, no ds_write, and a copy has two tile times to land (it is issued while tile i is multiplied and
// first read in tile i + 2).  The images are the unpadded [pixel][32 ch] 64-byte rows the transposing operand reads want, so a DMA wave-instruction simply
// fills 16 consecutive rows (lane l: row l / 4, piece l % 4); every wave issues the same number of pieces per tile (pieces beyond an image go, with
// out-of-range offsets, to a 1 KB strip nothing reads), so the wait before the tile's barrier is a counted vmcnt.  dbias: the column sums of dy are taken
// from the LDS image (the same 16-byte pieces in the same thread order as bwgrad2's staging registers: bit-identical).  Same products, same order: dw
// bit-identical to bwgrad2_kernel's.
template <int WCI, int WCO>
__global__ __launch_bounds__(512, 1) void bwgrad3_kernel(const BWgradParams p, const unsigned x_bytes, const unsigned dy_bytes) {
    constexpr int KS = 8 / (WCI * WCO), NSTEP = 8 / KS;
    constexpr int CIW = 32 * WCI, COW = 32 * WCO;
    constexpr int YQ = COW / 8;
    constexpr int YR = (WCO == 2) ? 2 : 1, NM = NSTEP * 9;
    constexpr int TP = 128, DYS = WCO * TP * 32;      // bf16 elements of the dy image
    constexpr int NDY = WCO;                           // dy pieces per wave: WCO * 8 pieces of 1 KB over 8 waves
    constexpr int NXM = 4;                             // x pieces per wave (<= 32 pieces: WCI * npix <= 512 rows, plan_bwgrad)
    constexpr int NL = NXM + NDY;
    static_assert(NL <= NM && NL <= 15, "one DMA per MFMA slot, counted vmcnt");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wci = wave % WCI, wco = (wave / WCI) % WCO, ks = wave / (WCI * WCO);
    int bid = blockIdx.x;
    const int split = bid % p.splits; bid /= p.splits;
    const int cob = bid % p.nCoB, cib = bid / p.nCoB;
    const int ci0 = cib * CIW, co0 = cob * COW;
    const int tinHW = p.TinH * p.TinW, npix = p.NB * tinHW;
    const int xrows = WCI * npix, xpieces = (xrows + 15) >> 4;
    const int XS = xpieces * 512;                      // bf16 elements of the x image, rounded up to whole 1 KB pieces
    const int STAGE = DYS + XS;                        // a stage: [dy image][x image]; three stages, then the 1 KB strip of the out-of-range pieces
    const unsigned lds_raw = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem_raw;
    const unsigned sink = lds_raw + 2u * 3u * (unsigned)STAGE;

    // lane constants of the transposing reads (as bwgrad2_kernel)
    const int g = lane >> 4, q = (lane & 15) >> 2, pc = lane & 3;
    const int chan = 16 * (g & 1) + 4 * pc, kbase = 8 * (g >> 1) + q;
    int xa0[NSTEP], xa1[NSTEP], ya0[NSTEP];
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) {
        const int m0 = 16 * (ks + s_ * KS) + kbase, m1 = m0 + 4;
        const int tx0 = m0 & (p.TW - 1), ty0 = (m0 >> p.lgTW) & (p.TH - 1), nb0 = m0 >> (p.lgTW + p.lgTH);
        const int tx1 = m1 & (p.TW - 1), ty1 = (m1 >> p.lgTW) & (p.TH - 1), nb1 = m1 >> (p.lgTW + p.lgTH);
        xa0[s_] = DYS + (wci * npix + (nb0 * p.TinH + ty0) * p.TinW + tx0) * 32 + chan;
        xa1[s_] = DYS + (wci * npix + (nb1 * p.TinH + ty1) * p.TinW + tx1) * 32 + chan;
        ya0[s_] = (wco * TP + m0) * 32 + chan;
    }
    const int rowp = p.TinW * 32;

    // DMA roles (tile-invariant): piece k = wave + 8 i of an image, lane l -> row 16 k + l / 4, 16-byte piece l % 4 (8 channels)
    int x_yx[NXM]; unsigned x_rel[NXM];               // x_yx = (nb << 16) | (iy << 8) | ix, or -1 (row beyond the image / channels beyond Ci)
#pragma unroll
    for (int i = 0; i < NXM; ++i) {
        const int k = wave + 8 * i, r = 16 * k + (lane >> 2), qq = lane & 3;
        const int grp = r / npix, pi = r - grp * npix;
        const int nb = pi / tinHW, rem = pi - nb * tinHW, iy = rem / p.TinW, ix = rem - iy * p.TinW;
        const bool on = k < xpieces && r < xrows && ci0 + 32 * grp + 8 * qq < p.Ci;
        x_yx[i] = on ? ((nb << 16) | (iy << 8) | ix) : -1;
        x_rel[i] = 2u * (unsigned)(((nb * p.H + iy) * p.W + ix) * p.ldx + ci0 + 32 * grp + 8 * qq);
    }
    int y_pos[NDY]; unsigned y_rel[NDY];
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
        const int k = wave + 8 * i, r = 16 * k + (lane >> 2), qq = lane & 3;      // k < WCO * 8 by construction
        const int grp = r / TP, m = r - grp * TP;
        const int tx = m & (p.TW - 1), ty = (m >> p.lgTW) & (p.TH - 1), nb = m >> (p.lgTW + p.lgTH);
        y_pos[i] = (co0 + 32 * grp + 8 * qq < p.Co) ? ((nb << 16) | (ty << 8) | tx) : -1;
        y_rel[i] = 2u * (unsigned)(((nb * p.H + ty) * p.W + tx) * p.lddy + co0 + 32 * grp + 8 * qq);
    }
    // dbias read-back roles: the pieces this thread staged in bwgrad2_kernel (idx = tid + it * 512 -> position idx / YQ, piece idx % YQ)
    int b_lds[YR];
#pragma unroll
    for (int it = 0; it < YR; ++it) {
        const int idx = tid + it * 512, m = idx / YQ, qq = idx - m * YQ;
        b_lds[it] = ((qq >> 2) * TP + m) * 32 + 8 * (qq & 3);
    }
    const bool want_bias = p.bias_slab != nullptr && cib == 0;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, dy_bytes, 0x00020000);

    // cursor of the tile whose copy is being issued (wave-uniform scalars)
    int ltile = split;
    bool c_live = false; int c_n0 = 0, c_h0 = 0, c_w0 = 0, c_a0 = 0, c_b0 = 0; unsigned c_xorg = 0, c_yorg = 0;
    auto next_tile = [&]() {
        c_live = ltile < p.tiles;
        int t_ = c_live ? ltile : 0;
        const int tb = t_ % p.tilesB; t_ /= p.tilesB;
        const int ta = t_ % p.tilesA;
        c_n0 = (t_ / p.tilesA) * p.NB; c_a0 = ta * p.TH; c_b0 = tb * p.TW;
        c_h0 = c_a0 + p.dh_min; c_w0 = c_b0 + p.dw_min;
        c_xorg = 2u * (unsigned)(((c_n0 * p.H + c_h0) * p.W + c_w0) * p.ldx);         // may wrap for halo origins: added mod 2^32
        c_yorg = 2u * (unsigned)(((c_n0 * p.H + c_a0) * p.W + c_b0) * p.lddy);
        ltile += p.splits;
    };
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(m0v), "s"(rs) : "memory");
    };
    auto dma1 = [&](int stage, int k_) {               // DMA k_ of NL of the cursor's tile into `stage`: x pieces first, then dy pieces
        if (k_ < NXM) {
            const int i = k_, k = wave + 8 * i;
            const int n = c_n0 + (x_yx[i] >> 16), h = c_h0 + ((x_yx[i] >> 8) & 255), w_ = c_w0 + (x_yx[i] & 255);
            const bool ok = c_live && x_yx[i] >= 0 && n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            const bool piece_ok = k < xpieces;          // wave-uniform
            dma(rs_x, (ok && piece_ok) ? c_xorg + x_rel[i] : BW_OOB, piece_ok ? lds_raw + 2u * (unsigned)(stage * STAGE + DYS) + 1024u * (unsigned)k : sink);
        } else {
            const int i = k_ - NXM, k = wave + 8 * i;
            const int n = c_n0 + (y_pos[i] >> 16), a = c_a0 + ((y_pos[i] >> 8) & 255), b = c_b0 + (y_pos[i] & 255);
            const bool ok = c_live && y_pos[i] >= 0 && n < p.N && a < p.H && b < p.W;
            dma(rs_dy, ok ? c_yorg + y_rel[i] : BW_OOB, lds_raw + 2u * (unsigned)(stage * STAGE) + 1024u * (unsigned)k);
        }
    };
    float bsum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] = 0.f;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- prologue: tiles 0 and 1 on their way into stages 0 and 1; tile 0 landed
    next_tile();
#pragma unroll
    for (int k = 0; k < NL; ++k) dma1(0, k);
    next_tile();
#pragma unroll
    for (int k = 0; k < NL; ++k) dma1(1, k);
    __builtin_amdgcn_s_waitcnt(0x0F70 | NL);
    __syncthreads();

    const int ntile = (p.tiles - split + p.splits - 1) / p.splits;
    auto iteration = [&](auto P_) {
        constexpr int P = decltype(P_)::value;          // stage of the tile being multiplied; its copy two tiles ahead goes into stage (P + 2) % 3
        const __bf16* sb = lds + bw_opaque(P * STAGE);
        next_tile();
        union Op { bf16x8 v; s16x4 h[2]; };
        constexpr int AD = BW2_AHEAD;
        Op aq[AD + 1], bq[2];
        const __bf16* r0 = sb; const __bf16* r1 = sb;
        auto read_a = [&](Op& o, int s_, int t) {
            if (t % 3 == 0) { r0 = sb + xa0[s_] + (t / 3) * rowp; r1 = sb + xa1[s_] + (t / 3) * rowp; }
            o.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(r0 + 32 * (t % 3)));
            o.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(r1 + 32 * (t % 3)));
        };
        auto read_b = [&](Op& o, int s_) {
            o.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sb + ya0[s_]));
            o.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sb + ya0[s_] + 128));
        };
        read_b(bq[0], 0);
#pragma unroll
        for (int k = 0; k < AD; ++k) read_a(aq[k], k / 9, k % 9);
        if (want_bias) {                               // column sums of this tile's dy image (block-uniform branch)
#pragma unroll
            for (int it = 0; it < YR; ++it) {
                union { bw_u32x4 u; bf16x8 v; } c;
                c.u = *reinterpret_cast<const bw_u32x4*>(sb + b_lds[it]);
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[j] += (float)c.v[j];
            }
        }
#pragma unroll
        for (int slot = 0; slot < NM; ++slot) {
            const int s_ = slot / 9, t = slot - 9 * s_;
            if (slot + AD < NM) {
                const int s1 = (slot + AD) / 9, t1 = (slot + AD) - 9 * s1;
                read_a(aq[(slot + AD) % (AD + 1)], s1, t1);
            }
            if (slot + 1 < NM && (slot + 1) % 9 == 0) read_b(bq[((slot + 1) / 9) & 1], (slot + 1) / 9);
            if (slot < NL) dma1((P + 2) % 3, slot);
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[slot % (AD + 1)].v, bq[s_ & 1].v, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70 | NL);       // all but the NL pieces just issued (tile i + 2) have landed: tile i + 1 is complete
        __syncthreads();
    };
    for (int i = 0; i < ntile; i += 3) {
        iteration(BwIC<0>{});
        if (i + 1 < ntile) iteration(BwIC<1>{});
        if (i + 2 < ntile) iteration(BwIC<2>{});
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                // the copies of the tiles past the end (zeros) must not land on the reduction buffer below
    __syncthreads();

    // ---- the KS waves of a channel block add their partials through LDS, then the ks = 0 wave writes the block (as bwgrad2_kernel)
    const int half = lane >> 5, e = lane & 31;
    const int co = co0 + 32 * wco + e;
    float* red_acc = reinterpret_cast<float*>(smem_raw);            // [8 waves][16][64]
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        if (ks > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red_acc[(wave * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (ks == 0 && co < p.Co) {
            float* dst = p.slab + (((long long)split * 9 + t) * p.Ci + ci0 + 32 * wci) * p.Co + co;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r];
#pragma unroll
                for (int k = 1; k < KS; ++k) v += red_acc[((wave + k * WCI * WCO) * 16 + r) * 64 + lane];
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                dst[(long long)row * p.Co] = v;
            }
        }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_raw);            // [512][8]
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bsum[k];
        __syncthreads();
        if (tid < COW) {
            const int qq = tid / 8, k = tid % 8;
            float sum = 0.f;
            for (int th = qq; th < 512; th += YQ) sum += red[th * 8 + k];
            if (co0 + tid < p.Co) p.bias_slab[(long long)split * p.Co + co0 + tid] = sum;
        }
    }
}

// out[i] = sum over s of slab[s][i]; the bias row likewise.  A block owns 64 consecutive outputs; its sixteen waves take the slabs
// s = w, w + 16, ... (each wave reads 256 contiguous bytes per slab) and meet in LDS: fixed order, bit-reproducible.  (One thread
// per output walking all S slabs serially was latency-bound: 60 us for the 18,432 outputs x 256 slabs of sp6.gamma+beta.)
struct BWTapMap { int cico; int t[16]; };            // cico > 0: tap k of the slab is tap t[k] of the output (stride-2 classes); 0: identity
__global__ __launch_bounds__(1024) void bwgrad_reduce_kernel(const float* __restrict__ slab, int S, long long n, float* __restrict__ out,
                                                             const float* __restrict__ bias_slab, int SB, int Co, float* __restrict__ dbias, int accumulate_bias,
                                                             const BWTapMap map) {
    __shared__ float red[16][64];
    const int o = threadIdx.x & 63, w = threadIdx.x >> 6;          // 16 waves: wave w takes the slabs w, w + 16, ... (four loads in flight)
    const long long nblk_w = (n + 63) / 64;
    const bool is_bias = (long long)blockIdx.x >= nblk_w;          // the trailing blocks sum the bias slabs the same way
    const float* src = is_bias ? bias_slab : slab;
    const long long cols = is_bias ? Co : n;
    const int rows = is_bias ? SB : S;
    const long long i = (is_bias ? (long long)blockIdx.x - nblk_w : (long long)blockIdx.x) * 64 + o;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < cols) {
        int k = w;
        for (; k + 48 < rows; k += 64) {
            s0 += src[(long long)k * cols + i]; s1 += src[(long long)(k + 16) * cols + i];
            s2 += src[(long long)(k + 32) * cols + i]; s3 += src[(long long)(k + 48) * cols + i];
        }
        for (; k < rows; k += 16) s0 += src[(long long)k * cols + i];
    }
    red[w][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && i < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][o];
        if (is_bias) dbias[i] = accumulate_bias ? dbias[i] + t : t;
        else if (map.cico > 0) { const int k = (int)(i / map.cico); out[(long long)map.t[k] * map.cico + (i - (long long)k * map.cico)] = t; }
        else out[i] = t;
    }
}

struct BWgradPlan { BWgradParams p; int wci, wco, KS; size_t lds; long long slab_floats, bias_floats; };
struct BWgradPlanS2 { BWgradPack pk; BWTapMap map; int wci, wco, grid; size_t lds; long long slab_floats, bias_floats; };

// stride 2, pad 1, 3x3 / 4x4, even H and W, bf16 views: the four parity classes (comment at bwgrad_pack_kernel)
static int plan_bwgrad_s2(BWgradPlanS2& pl, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad) {
    if (stride != 2 || pad != 1 || kh != kw || (kh != 3 && kh != 4) || (H & 1) || (W & 1)) return MRDIS_EUNSUPPORTED;
    if ((Ci % 32 != 0 && Ci != 16) || Co % 8 != 0 || Co < 16) return MRDIS_EUNSUPPORTED;
    const int Hc = H / 2, Wc = W / 2;                               // class view = output map size
    if ((H + 2 * pad - kh) / 2 + 1 != Hc || (W + 2 * pad - kw) / 2 + 1 != Wc) return MRDIS_EUNSUPPORTED;
    if ((long long)N * Hc * Wc < 2048) return MRDIS_EUNSUPPORTED;                                  // tiny maps: the fp32 kernels' slabs are cheaper
    int tw = 32; while (tw > 1 && tw / 2 >= Wc) tw >>= 1;
    int th = 128 / tw; while (th > 1 && th / 2 >= Hc) th >>= 1;
    const int nb = 128 / (tw * th);
    if (nb > 255) return MRDIS_EUNSUPPORTED;
    pl.wci = (Ci % 64 == 0) ? 2 : 1;
    pl.wco = (Co > 32) ? 2 : 1;
    const int TT = kh * kw;
    pl.map = BWTapMap{};
    pl.map.cico = Ci * Co;
    pl.lds = 8 * 16 * 64 * 4;                                        // the epilogue's wave-reduction buffer / bias scratch
    int tbase = 0;
    long long tiles = 0;
    for (int cls = 0; cls < 4; ++cls) {
        const int pr = cls >> 1, pq = cls & 1;
        BWgradParams& p = pl.pk.c[cls];
        p = BWgradParams{};
        p.N = N; p.H = Hc; p.W = Wc; p.Ci = Ci; p.Co = Co;
        int dh_min = 9, dh_max = -9, dw_min = 9, dw_max = -9;
        for (int r = 0; r < kh; ++r) for (int s_ = 0; s_ < kw; ++s_) {
            if (((r - pad) & 1) != pr || ((s_ - pad) & 1) != pq) continue;
            const int dh = (r - pad) >> 1, dw = (s_ - pad) >> 1;           // arithmetic shift: floor
            p.dh[p.ntaps] = dh; p.dw[p.ntaps] = dw; pl.map.t[tbase + p.ntaps] = r * kw + s_; ++p.ntaps;
            dh_min = dh < dh_min ? dh : dh_min; dh_max = dh > dh_max ? dh : dh_max;
            dw_min = dw < dw_min ? dw : dw_min; dw_max = dw > dw_max ? dw : dw_max;
        }
        if (p.ntaps < 1) return MRDIS_EUNSUPPORTED;
        p.dh_min = dh_min; p.dw_min = dw_min;
        p.TW = tw; p.TH = th; p.NB = nb;
        p.lgTW = 0; while ((1 << p.lgTW) < tw) ++p.lgTW;
        p.lgTH = 0; while ((1 << p.lgTH) < th) ++p.lgTH;
        p.TinH = th + dh_max - dh_min; p.TinW = tw + dw_max - dw_min;
        p.tilesA = mrdis_cdiv(Hc, th); p.tilesB = mrdis_cdiv(Wc, tw); p.tilesN = mrdis_cdiv(N, nb);
        tiles = (long long)p.tilesA * p.tilesB * p.tilesN;
        if (tiles > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
        p.tiles = (int)tiles;
        p.nCiB = mrdis_cdiv(Ci, 32 * pl.wci); p.nCoB = mrdis_cdiv(Co, 32 * pl.wco);
        const long long npix = (long long)nb * p.TinH * p.TinW;
        if (npix * (4 * pl.wci) > (long long)(pl.wci == 2 ? 4 : 2) * 512) return MRDIS_EUNSUPPORTED;
        const size_t lds = 2 * 32 * ((size_t)pl.wco * 128 + (size_t)pl.wci * npix);
        if (lds > pl.lds) pl.lds = lds;
        p.tstride = TT; p.tbase = tbase;
        tbase += p.ntaps;
    }
    if (tbase != TT) return MRDIS_EUNSUPPORTED;
    // the four classes together fill the chip: a quarter of the workgroups each
    long long splits = mrdis_cdiv(bconv_ncu(), 4LL * pl.pk.c[0].nCiB * pl.pk.c[0].nCoB);
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    for (int cls = 0; cls < 4; ++cls) pl.pk.c[cls].splits = (int)splits;
    pl.grid = (int)splits * pl.pk.c[0].nCiB * pl.pk.c[0].nCoB;
    pl.slab_floats = splits * TT * Ci * Co;
    pl.bias_floats = splits * Co;
    return MRDIS_OK;
}

static int run_bwgrad_s2(const void* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                         int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int accumulate_bias, hipStream_t s) {
    BWgradPlanS2 pl;
    int rc = plan_bwgrad_s2(pl, N, H, W, Ci, Co, kh, kw, stride, pad);
    if (rc) return rc;
    if (ldx % 8 != 0 || lddy % 8 != 0 || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)workspace) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)H * W * ldx > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes < sizeof(float) * (size_t)(pl.slab_floats + pl.bias_floats) + 256) return MRDIS_EWORKSPACE;
    float* slab = reinterpret_cast<float*>(workspace);
    for (int cls = 0; cls < 4; ++cls) {
        BWgradParams& p = pl.pk.c[cls];
        p.x = reinterpret_cast<const __bf16*>(x) + ((long long)(cls >> 1) * W + (cls & 1)) * ldx;       // x[:, p::2, q::2]
        p.dy = dy; p.ldx = 2 * ldx; p.lddy = lddy;
        p.rsx = 2 * W * ldx; p.isx = H * W * ldx;
        p.slab = slab;
        p.bias_slab = (dbias && cls == 0) ? slab + pl.slab_floats : nullptr;       // every class walks all of dy: one of them sums its columns
    }
#define BWP_CASE(a, b_, HF) if (pl.wci == a && pl.wco == b_ && (Ci == 16) == HF) { \
        static bool attr_set = false; \
        if (!attr_set) { if (hipFuncSetAttribute((const void*)bwgrad_pack_kernel<a, b_, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return MRDIS_ELAUNCH; attr_set = true; } \
        MRDIS_LAUNCH((bwgrad_pack_kernel<a, b_, HF>), dim3(pl.grid, 4), dim3(512), pl.lds, s, pl.pk); }
    BWP_CASE(1, 1, true) else BWP_CASE(1, 2, true) else BWP_CASE(1, 1, false) else BWP_CASE(1, 2, false) else BWP_CASE(2, 1, false) else BWP_CASE(2, 2, false)
    else return MRDIS_EUNSUPPORTED;
#undef BWP_CASE
    MRDIS_CHECK_LAUNCH();
    const long long n = (long long)kh * kw * Ci * Co;
    const int splits = pl.pk.c[0].splits;
    MRDIS_LAUNCH(bwgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64 + (dbias ? (Co + 63) / 64 : 0))), dim3(1024), 0, s, slab, splits, n, dw_tck,
                       slab + pl.slab_floats, splits, Co, dbias, accumulate_bias, pl.map);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

static int plan_bwgrad(BWgradPlan& pl, int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int dtype = MRDIS_DT_F32_BF16M) {
    if (stride != 1 || kh * kw > 9 || kh != kw || 2 * pad != kh - 1) return MRDIS_EUNSUPPORTED;    // "same" convolutions only (Ho = H)
    if ((Ci % 32 != 0 && !(Ci == 16 && dtype == MRDIS_DT_BF16)) || Co % 8 != 0 || Co < 16) return MRDIS_EUNSUPPORTED;   // Ci = 16: HALF instantiation
    // measured (tools/layer_bench.py --dtype bf16, B = 32): with 32 or fewer couts a wave's 32 x 32 block leaves too few waves per
    // channel block and the fp32 narrow-cout kernels win (sp6.out 593 vs 249 us, sp5.out 310 vs 198 us) unless Cin >= 128
    // (with bf16 activations the alternative is two full-size view casts in front of the fp32 kernel: stay here)
    if (Co <= 32 && Ci < 128 && dtype != MRDIS_DT_BF16) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W < 4096) return MRDIS_EUNSUPPORTED;                                    // tiny maps: the fp32 kernels' slabs are cheaper
    BWgradParams& p = pl.p;
    p = BWgradParams{};
    p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co;
    p.ntaps = kh * kw;
    for (int r = 0; r < kh; ++r) for (int s_ = 0; s_ < kw; ++s_) { p.dh[r * kw + s_] = r - pad; p.dw[r * kw + s_] = s_ - pad; }
    p.dh_min = -pad; p.dw_min = -pad;
    // 128-position tile with power-of-two sides
    int tw = 32; while (tw > 1 && tw / 2 >= W) tw >>= 1;
    int th = 128 / tw; while (th > 1 && th / 2 >= H) th >>= 1;
    int nb = 128 / (tw * th);
    p.TW = tw; p.TH = th; p.NB = nb;
    p.lgTW = 0; while ((1 << p.lgTW) < tw) ++p.lgTW;
    p.lgTH = 0; while ((1 << p.lgTH) < th) ++p.lgTH;
    p.TinH = th + kh - 1; p.TinW = tw + kw - 1;
    if (p.TinH > 255 || p.TinW > 255 || nb > 255) return MRDIS_EUNSUPPORTED;
    p.tilesA = mrdis_cdiv(H, th); p.tilesB = mrdis_cdiv(W, tw); p.tilesN = mrdis_cdiv(N, nb);
    const long long tiles = (long long)p.tilesA * p.tilesB * p.tilesN;
    if (tiles > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.tiles = (int)tiles;
    pl.wci = (Ci % 64 == 0) ? 2 : 1;
    pl.wco = (Co > 32) ? 2 : 1;
    pl.KS = 8 / (pl.wci * pl.wco);
    p.nCiB = mrdis_cdiv(Ci, 32 * pl.wci); p.nCoB = mrdis_cdiv(Co, 32 * pl.wco);
    const long long npix = (long long)nb * p.TinH * p.TinW;
    if (npix * (4 * pl.wci) > (long long)(pl.wci == 2 ? 4 : 2) * 512) return MRDIS_EUNSUPPORTED;
    pl.lds = 2 * 32 * ((size_t)pl.wco * 128 + (size_t)pl.wci * npix);
    if (pl.lds < 8 * 16 * 64 * 4) pl.lds = 8 * 16 * 64 * 4;          // the epilogue's wave-reduction buffer / bias scratch
    long long splits = mrdis_cdiv(bconv_ncu(), (long long)p.nCiB * p.nCoB);
    if (splits > tiles) splits = tiles;
    if (splits < 1) splits = 1;
    p.splits = (int)splits;
    pl.slab_floats = (long long)p.splits * p.ntaps * Ci * Co;
    pl.bias_floats = (long long)p.splits * Co;
    p.tstride = p.ntaps; p.tbase = 0;
    return MRDIS_OK;
}

size_t mrdis_bwgrad_workspace(int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad) {
    if (stride == 2) {
        BWgradPlanS2 p2;
        if (plan_bwgrad_s2(p2, N, H, W, Ci, Co, kh, kw, stride, pad)) return 0;
        return sizeof(float) * (size_t)(p2.slab_floats + p2.bias_floats) + 256;
    }
    BWgradPlan pl;
    if (plan_bwgrad(pl, N, H, W, Ci, Co, kh, kw, stride, pad, MRDIS_DT_BF16)) return 0;          // the widest domain
    return sizeof(float) * (size_t)(pl.slab_floats + pl.bias_floats) + 256;
}

int mrdis_run_bwgrad(const void* x, int ldx, const void* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                     int N, int H, int W, int Ci, int Co, int kh, int kw, int stride, int pad, int accumulate_bias, int dtype, hipStream_t s) {
    const bool st_bf16 = dtype == MRDIS_DT_BF16;
    if (stride == 2) {
        if (!st_bf16 || mrdis_opt(MRDIS_OPT_NOW16)) return MRDIS_EUNSUPPORTED;      // fp32 views: the fp32 parity-class kernels (mrdis_conv.hip)
        return run_bwgrad_s2(x, ldx, dy, lddy, dw_tck, dbias, workspace, workspace_bytes, N, H, W, Ci, Co, kh, kw, stride, pad, accumulate_bias, s);
    }
    BWgradPlan pl;
    int rc = plan_bwgrad(pl, N, H, W, Ci, Co, kh, kw, stride, pad, dtype);
    if (rc) return rc;
    if (ldx % (st_bf16 ? 8 : 4) != 0 || lddy % (st_bf16 ? 8 : 4) != 0 || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)workspace) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes < sizeof(float) * (size_t)(pl.slab_floats + pl.bias_floats) + 256) return MRDIS_EWORKSPACE;
    BWgradParams& p = pl.p;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy;
    if ((long long)H * W * ldx > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.rsx = W * ldx; p.isx = H * W * ldx;
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + pl.slab_floats : nullptr;
    const int grid = p.splits * p.nCiB * p.nCoB;
#define BW_CASE_H(a, b_, TS, HF) if (pl.wci == a && pl.wco == b_) { \
        static bool attr_set = false; \
        if (!attr_set) { if (hipFuncSetAttribute((const void*)bwgrad_kernel<a, b_, TS, HF>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return MRDIS_ELAUNCH; attr_set = true; } \
        MRDIS_LAUNCH((bwgrad_kernel<a, b_, TS, HF>), dim3(grid), dim3(512), pl.lds, s, p); }
#define BW_CASE(a, b_, TS) BW_CASE_H(a, b_, TS, false)
    bool done2 = false;
    if (st_bf16 && Ci % 32 == 0 && mrdis_opt(MRDIS_OPT_WINO_PIPE)) {              // pipelined form (bwgrad2_kernel); needs both images twice in LDS
        const long long xb = 2LL * (((long long)N * H * W - 1) * ldx + Ci), yb = 2LL * (((long long)N * H * W - 1) * lddy + Co);
        const size_t lds2 = 2 * (size_t)pl.lds;
        bool canon = p.ntaps == 9;                    // tap t = row t / 3, column t % 3 of the 3x3 window (what the kernel's addressing assumes)
        for (int t_ = 0; canon && t_ < 9; ++t_) canon = (p.dh[t_] - p.dh_min == t_ / 3) && (p.dw[t_] - p.dw_min == t_ % 3);
        // round 5: the LDS-DMA form (three stages + a 1 KB strip); debug_mode 3010 keeps bwgrad2_kernel (A/B)
        const long long npix3 = (long long)p.NB * p.TinH * p.TinW;
        const size_t stage3 = 2 * ((size_t)pl.wco * 128 * 32 + (size_t)(((pl.wci * npix3 + 15) / 16) * 512));
        const size_t lds3 = 3 * stage3 + 1024 < 32768 ? 32768 : 3 * stage3 + 1024;
        if (canon && xb < 0xffffffe0LL && yb < 0xffffffe0LL && lds3 <= 160 * 1024 && pl.wci * npix3 <= 512 && p.tiles >= 3 * p.splits && mrdis_opt(MRDIS_OPT_MODE) != 3010) {
#define BW3_CASE(a, b_) if (pl.wci == a && pl.wco == b_) { \
            static bool attr3 = false; \
            if (!attr3) { if (hipFuncSetAttribute((const void*)bwgrad3_kernel<a, b_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MRDIS_ELAUNCH; attr3 = true; } \
            MRDIS_LAUNCH((bwgrad3_kernel<a, b_>), dim3(grid), dim3(512), lds3, s, p, (unsigned)xb, (unsigned)yb); done2 = true; }
            BW3_CASE(1, 1) else BW3_CASE(1, 2) else BW3_CASE(2, 1) else BW3_CASE(2, 2)
#undef BW3_CASE
        }
        if (!done2 && canon && xb < 0xffffffe0LL && yb < 0xffffffe0LL && lds2 <= 150 * 1024 && p.tiles >= 2 * p.splits) {
#define BW2_CASE(a, b_) if (pl.wci == a && pl.wco == b_) { \
            static bool attr2 = false; \
            if (!attr2) { if (hipFuncSetAttribute((const void*)bwgrad2_kernel<a, b_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return MRDIS_ELAUNCH; attr2 = true; } \
            MRDIS_LAUNCH((bwgrad2_kernel<a, b_>), dim3(grid), dim3(512), lds2 < 32768 ? 32768 : lds2, s, p, (unsigned)xb, (unsigned)yb); done2 = true; }
            BW2_CASE(1, 1) else BW2_CASE(1, 2) else BW2_CASE(2, 1) else BW2_CASE(2, 2)
#undef BW2_CASE
        }
    }
    if (done2) {}
    else if (Ci == 16) { BW_CASE_H(1, 1, __bf16, true) else BW_CASE_H(1, 2, __bf16, true) }
    else if (st_bf16) { BW_CASE(1, 1, __bf16) else BW_CASE(1, 2, __bf16) else BW_CASE(2, 1, __bf16) else BW_CASE(2, 2, __bf16) }
    else { BW_CASE(1, 1, float) else BW_CASE(1, 2, float) else BW_CASE(2, 1, float) else BW_CASE(2, 2, float) }
#undef BW_CASE
#undef BW_CASE_H
    MRDIS_CHECK_LAUNCH();
    const long long n = (long long)p.ntaps * Ci * Co;
    MRDIS_LAUNCH(bwgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64 + (dbias ? (Co + 63) / 64 : 0))), dim3(1024), 0, s, p.slab, p.splits, n, dw_tck,
                       p.bias_slab, p.splits, Co, dbias, accumulate_bias, BWTapMap{});
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// launch what mrdis_run_bconv planned for the four parity classes: one launch when they share the instantiation, else one each
int mrdis_launch_bconv_planned(const BConvLaunch (&L)[4], hipStream_t s) {
    bool same = true;
    for (int k = 0; k < 4; ++k) same = same && L[k].set;
    for (int k = 1; k < 4 && same; ++k)
        same = L[k].KC == L[0].KC && L[k].waves_c == L[0].waves_c && L[k].wp == L[0].wp && L[k].wc == L[0].wc && L[k].p.dtype == L[0].p.dtype;
    if (same) {
#define BP_CASE(kc, a, b_, d) if (L[0].KC == kc && L[0].waves_c == a && L[0].wp == b_ && L[0].wc == d) \
        return L[0].p.dtype == MRDIS_DT_BF16 ? launch_bconv_pack_t<kc, a, b_, d, __bf16>(L, s) : launch_bconv_pack_t<kc, a, b_, d, float>(L, s)
        BP_CASE(32, 1, 2, 2); BP_CASE(16, 1, 2, 2); BP_CASE(32, 1, 1, 2); BP_CASE(16, 1, 1, 2);
        BP_CASE(32, 1, 2, 1); BP_CASE(16, 1, 2, 1); BP_CASE(32, 1, 1, 1); BP_CASE(16, 1, 1, 1);
#undef BP_CASE
        return MRDIS_EUNSUPPORTED;
    }
    for (int k = 0; k < 4; ++k) {
        if (!L[k].set) continue;
        int rc = MRDIS_EUNSUPPORTED;
#define B1_CASE(kc, a, b_, d) if (L[k].KC == kc && L[k].waves_c == a && L[k].wp == b_ && L[k].wc == d) rc = launch_bconv<kc, a, b_, d>(L[k].p, L[k].g, L[k].grid, L[k].lds, s)
        B1_CASE(32, 1, 2, 2); B1_CASE(16, 1, 2, 2); B1_CASE(32, 1, 1, 2); B1_CASE(16, 1, 1, 2);
        B1_CASE(32, 1, 2, 1); B1_CASE(16, 1, 2, 1); B1_CASE(32, 1, 1, 1); B1_CASE(16, 1, 1, 1);
#undef B1_CASE
        if (rc) return rc;
    }
    return MRDIS_OK;
}
