// mrdis_wino.hip -- fused Winograd F(2x2, 3x3) convolution for gfx950 (MI355X), fp32, NHWC.
//
// The 3x3 / stride 1 / pad 1 layers are 95 % of the step's FLOPs.  F(2x2, 3x3) computes a 2x2
// output tile from a 4x4 input tile with 16 multiplies per (ci, co) instead of 36:
//
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A          (Lavin & Gray 2016; the transforms are exact in binary fp: +-, x0.5)
//
// so the MFMA work drops 2.25x; the price is the two transforms, which run on the VALU out of LDS.  Everything is
// fused in ONE kernel (no transformed tensors in HBM, same C ABI as the direct kernel):
//
//   workgroup (8 waves) = T tiles (T = 64: 8x8 tiles = 16x16 outputs, or T = 128: 8x16 tiles) x BN couts (64 or 32)
//   per 8-channel chunk:
//     1. the halo'd raw input block ((2TBH+2) x (2TBW+2) pixels x 8 ch) and the 9 filter taps of each (ci, co) pair of
//        the chunk were prefetched into registers during the previous chunk's MFMAs; they go to LDS -- the filter
//        THROUGH G g G^T, i.e. as U[xi][ci][co], xi = 0..15
//     2. every thread transforms tile-channels: 16 raw reads -> B^T d B -> V[xi][ci][tile]
//     3. 16 independent products D_xi[co][tile] += U_xi[co][ci] V_xi[ci][tile] on v_mfma_f32_16x16x4_f32: a wave owns
//        16 tiles x 32 couts for ALL 16 xi (128 accumulator registers), so the output transform is lane-local
//   epilogue: Y = A^T M A per lane (24 adds per 2x2 tile and cout), + bias (+ LeakyReLU), 16-byte stores.
//
// The data gradient of such a layer is the same convolution with the taps reversed and [tap][Co][Ci] filters.
// Results differ from the direct kernel by fp32 rounding of the transforms (~1e-6 relative; parity bar 1e-3).
#include "mrdis_common.h"
#include "mrdis_wino.h"
#include <stdlib.h>

struct WinoParams {
    const float* in; const float* w; const float* bias; float* out;
    const float* res; int ldres;   // D3: residual added in the epilogue (BasicBlock), or nullptr
    int D;                        // D3: planes per sample; N counts output planes (samples x D), the filter has 27 taps
    int N, H, W, Cin, ldin, Cout, ldout;
    int flip, lrelu, nt_out;      // nt_out: output stream larger than the caches can keep for its consumer -> non-temporal stores
    int TBH, TBW;                 // tile-block shape in tiles
    int nby, nbx, coTiles;        // tile blocks per image, cout tiles
};

// CG cout groups of 32 x TG tile groups of 16 = waves per workgroup, all on 8x8 tile blocks (16x16 outputs):
//   <2,4>: 8 waves, 64 tiles x 64 couts, one workgroup per CU (Cout > 32)
//   <1,4>: 4 waves, 64 tiles x 32 couts, two workgroups per CU (Cout <= 32): one transforms while the other multiplies
// Timing-only builds (-DWINO_ABL_NOV / _NOU / _NOMFMA) on a 128 -> 128 layer at 64x64, B = 32: 218 us = 103 us of
// MFMA (the matrix pipe at its peak rate) + 40 us V transform + 28 us U transform + 47 us staging, barriers,
// epilogue -- the phases of one workgroup run back to back.  Splitting the 64-cout workgroup into two 4-wave ones
// (32 tiles x 64 couts, or 64 tiles x 32 couts) to overlap them doubles one of the transforms and was slower
// (243 us); for Cout = 32 nothing is duplicated and the overlap is worth 6 %.  Transforming the filters once per call
// in a pre-pass (16 loads per pair instead of 9 + G g G^T) was not faster either: the U phase is its loads and LDS
// writes, not the arithmetic.  Two barriers per chunk instead of three (raw block of chunk c+1 written during the
// MFMAs of chunk c) was 8 % slower (tools/ab_lib.py, same process).  Wave specialisation -- 8 multiplying waves + 4
// transforming waves per workgroup, double-buffered XOR-swizzled U / V, producers fed straight from global memory,
// one LDS-only barrier per chunk -- was parity-green but 25 % slower (283 us): the consumers alone took 148 us, the
// producers alone 167 us, and together they ran almost back to back instead of side by side.
// D3: the 3x3x3 layers of the 3-D nets as a hybrid -- Winograd F(2x2,3x3) in (h, w), direct in depth: an output plane d is
// sum over kd of the 2-D Winograd convolution of input plane d + kd - 1 with the 3x3 slice kd of the filter, i.e. the same
// kernel with (kd, ci) as the reduction axis (a chunk whose plane lies outside the volume contributes zeros).
template <int CG, int TG, bool D3 = false>
__global__ __launch_bounds__(64 * CG * TG, (CG * TG == 8) ? 1 : 2) void wino_conv_kernel(const WinoParams p) {
    constexpr int KC = 8, NT = 64 * CG * TG;
    constexpr int RWP = 24;                           // row pitch of a raw channel plane (18 pixels used): see the V transform below
    constexpr int BN = 32 * CG, T = 16 * TG;
    constexpr int TS = T + 16, US = BN + 16;          // row pitches: the 4 k-rows of an MFMA operand land 16 banks apart
    constexpr int XR = NT == 512 ? 2 : 3;             // raw float4 items per thread (host: RH * RW * 2 <= XR * NT)
    constexpr int UR = (KC * BN) / NT;                // filter (ci, co) pairs per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* U = smem;                                  // [16][KC][US]
    float* V = U + 16 * KC * US;                      // [16][KC][TS]
    float* raw = V + 16 * KC * TS;                    // [KC][RH][RWP]: one plane per channel of the chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int cg = wave % CG, tg = wave / CG;
    const int RH = 2 * p.TBH + 2, RW = 2 * p.TBW + 2, npix = RH * RW;
    const int PIXP = RH * RWP;                        // floats per raw channel plane

    int bid = mrdis_xcd_remap(blockIdx.x, gridDim.x);
    const int cot = bid % p.coTiles; bid /= p.coTiles;
    const int bx = bid % p.nbx; bid /= p.nbx;
    const int by = bid % p.nby;
    const int n = bid / p.nby;
    const int co0 = cot * BN;
    const int oy0 = 2 * by * p.TBH, ox0 = 2 * bx * p.TBW;      // output origin of the block; raw origin is (oy0 - 1, ox0 - 1)

    // ---- chunk-invariant staging roles
    int xg[XR], xl[XR];
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * NT;
        xg[it] = -1; xl[it] = -1;
        if (idx < npix * 2) {
            const int pi = idx >> 1, q = idx & 1;
            const int ry = pi / RW, rx = pi - ry * RW;
            const int h = oy0 - 1 + ry, w_ = ox0 - 1 + rx;
            xl[it] = 4 * q * PIXP + ry * RWP + rx;       // channel 4q of the pixel; its three neighbours are PIXP apart
            if ((unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                xg[it] = ((n * p.H + h) * p.W + w_) * p.ldin + 4 * q;            // host: < 2^31 elements
        }
    }
    const int fk = tid / BN, fco = tid - fk * BN;                                   // filter pairs of this thread: (fk + u * NT / BN, fco)
    const bool f_on = (co0 + fco) < p.Cout;
    float4 xr[XR];
    float gr[UR][9];
    const int nch = (p.Cin + KC - 1) / KC, nchunks = D3 ? 3 * nch : nch;
    const int dpl = D3 ? n % p.D : 0;                  // depth of this workgroup's output plane
    const int plane_pitch = p.H * p.W * p.ldin;
    auto load_chunk = [&](int cc) {
        const int kd = D3 ? cc / nch : 1;
        const int c0 = (D3 ? cc - kd * nch : cc) * KC;
        const bool plane_ok = !D3 || (unsigned)(dpl + kd - 1) < (unsigned)p.D;
        const int xshift = D3 ? (kd - 1) * plane_pitch : 0;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int q4 = 4 * ((tid + it * NT) & 1);
            if (plane_ok && xg[it] >= 0 && c0 + q4 < p.Cin) xr[it] = *reinterpret_cast<const float4*>(p.in + xg[it] + xshift + c0);
        }
        const long long tstride = (long long)p.Cin * p.Cout;
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int k = fk + u * (NT / BN);
            const bool on = f_on && plane_ok && (c0 + k) < p.Cin;
            const float* wp = p.w + ((long long)(c0 + k)) * p.Cout + co0 + fco;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ti = D3 ? kd * 9 + t : t;
                gr[u][t] = on ? wp[(p.flip ? (D3 ? 26 : 8) - ti : ti) * tstride] : 0.f;
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < XR; ++it)
            if (xl[it] >= 0) { float* d = raw + xl[it]; d[0] = xr[it].x; d[PIXP] = xr[it].y; d[2 * PIXP] = xr[it].z; d[3 * PIXP] = xr[it].w; }
#ifndef WINO_ABL_NOU
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            // U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
            float t_[4][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float g0 = gr[u][j], g1 = gr[u][3 + j], g2 = gr[u][6 + j];
                t_[0][j] = g0; t_[1][j] = 0.5f * (g0 + g1 + g2); t_[2][j] = 0.5f * (g0 - g1 + g2); t_[3][j] = g2;
            }
            float* up = U + (fk + u * (NT / BN)) * US + fco;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float u0 = t_[a][0], u3 = t_[a][2];
                const float u1 = 0.5f * (t_[a][0] + t_[a][1] + t_[a][2]), u2 = 0.5f * (t_[a][0] - t_[a][1] + t_[a][2]);
                up[(4 * a + 0) * KC * US] = u0; up[(4 * a + 1) * KC * US] = u1;
                up[(4 * a + 2) * KC * US] = u2; up[(4 * a + 3) * KC * US] = u3;
            }
        }
#endif
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int a_off = kq * US + 32 * cg + l16;        // + (xi * KC + 4 * ks) * US (+ 16 for the second block)
    const int b_off = kq * TS + 16 * tg + l16;        // + (xi * KC + 4 * ks) * TS

    load_chunk(0);
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc) __syncthreads();                      // the previous chunk's MFMAs have read U / V
        store_chunk();
        __syncthreads();
#ifndef WINO_ABL_NOV
        // ---- V = B^T d B per (tile, channel): B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
        for (int rep = 0; rep < (T * KC) / NT; ++rep) {
            const int tc = tid + rep * NT;
            const int tile = tc % T, ch = tc / T;
            const int ty = tile / p.TBW, tx = tile - ty * p.TBW;
            // Raw block as channel planes with a 24-float row pitch, read as two 8-byte pairs per row.  The 32 lanes of a read
            // group are the tiles (tx 0..7, ty 0..3) of one channel: 2 tx covers 16 consecutive banks and 2 ty * 24 = 48 ty steps
            // the four rows to distinct quarters of the 64 banks -- conflict-free.  (The former [pixel][9] image read 16 single
            // floats at 18 tx + 324 ty: 18 tx + 4 ty mod 32 collides up to 4-way; PMC: 1.17 conflict cycles per LDS instruction.)
            const float* rp = raw + ch * PIXP + (2 * ty) * RWP + 2 * tx;
            float d[4][4], r[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float2 lo = *reinterpret_cast<const float2*>(rp + i * RWP), hi = *reinterpret_cast<const float2*>(rp + i * RWP + 2);
                d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[0][j] = d[0][j] - d[2][j]; r[1][j] = d[1][j] + d[2][j];
                r[2][j] = d[2][j] - d[1][j]; r[3][j] = d[1][j] - d[3][j];
            }
            float* vp = V + ch * TS + tile;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                vp[(4 * a + 0) * KC * TS] = r[a][0] - r[a][2];
                vp[(4 * a + 1) * KC * TS] = r[a][1] + r[a][2];
                vp[(4 * a + 2) * KC * TS] = r[a][2] - r[a][1];
                vp[(4 * a + 3) * KC * TS] = r[a][1] - r[a][3];
            }
        }
#endif
        __syncthreads();
        if (cc + 1 < nchunks) load_chunk(cc + 1);      // global loads in flight during the MFMAs
#ifndef WINO_ABL_NOMFMA
        // 32 steps (xi, 4-channel group), operands fetched two steps ahead of the MFMAs that use them (three rotating
        // register sets; the sched_barriers keep hipcc from sinking the reads back next to their use -- left alone it
        // emits read, wait, 2 MFMAs, wait, 2 MFMAs and every pair of MFMAs eats a full LDS round trip)
        {
            constexpr int NS = 16 * (KC / 4);
            float a0[3], a1[3], bv[3];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int row = ((s_ / (KC / 4)) * KC + 4 * (s_ % (KC / 4)));
                a0[s_] = U[row * US + a_off]; a1[s_] = U[row * US + a_off + 16]; bv[s_] = V[row * TS + b_off];
            }
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                if (s_ + 2 < NS) {
                    const int row = (((s_ + 2) / (KC / 4)) * KC + 4 * ((s_ + 2) % (KC / 4)));
                    a0[(s_ + 2) % 3] = U[row * US + a_off]; a1[(s_ + 2) % 3] = U[row * US + a_off + 16]; bv[(s_ + 2) % 3] = V[row * TS + b_off];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int x = s_ / (KC / 4);
                acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s_ % 3], bv[s_ % 3], acc[x][0], 0, 0, 0);
                acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s_ % 3], bv[s_ % 3], acc[x][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
    }

    // ---- epilogue: lane = tile (16 * tg + l16), couts co0 + 32 cg + 16 b + 4 kq + r; Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]
    const int tile = 16 * tg + l16;
    const int ty = tile / p.TBW, tx = tile - ty * p.TBW;
    const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
    if (ty >= p.TBH || oy >= p.H || ox >= p.W) return;
    const bool vec_out = (p.ldout % 4 == 0) && (((uintptr_t)p.out & 15) == 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int co = co0 + 32 * cg + 16 * b + 4 * kq;
        if (co >= p.Cout) continue;
        float y[2][2][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t0[4], t1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = acc[4 * i][b][r], m1 = acc[4 * i + 1][b][r], m2 = acc[4 * i + 2][b][r], m3 = acc[4 * i + 3][b][r];
                t0[i] = m0 + m1 + m2; t1[i] = m1 - m2 - m3;
            }
            y[0][0][r] = t0[0] + t0[1] + t0[2]; y[0][1][r] = t1[0] + t1[1] + t1[2];
            y[1][0][r] = t0[1] - t0[2] - t0[3]; y[1][1][r] = t1[1] - t1[2] - t1[3];
        }
        float bq[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (co + r < p.Cout) bq[r] = p.bias[co + r];
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                if (oy + dy >= p.H || ox + dx >= p.W) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = y[dy][dx][r] + bq[r];
                    if (p.lrelu) v[r] = v[r] > 0.f ? v[r] : 0.2f * v[r];
                }
                if (D3 && p.res != nullptr) {
                    const float* rs = p.res + ((long long)(n * p.H + oy + dy) * p.W + ox + dx) * p.ldres + co;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (co + r < p.Cout) v[r] += rs[r];
                }
                float* dst = p.out + ((long long)(n * p.H + oy + dy) * p.W + ox + dx) * p.ldout + co;
                if (vec_out && co + 3 < p.Cout) {
                    if (p.nt_out) __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(dst));
                    else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (co + r < p.Cout) dst[r] = v[r];
                }
            }
    }
}

template <int CG, int TG>
static size_t wino_lds(int TBH, int TBW) {
    constexpr int KC = 8, BN = 32 * CG, T = 16 * TG;
    return sizeof(float) * ((size_t)16 * KC * (BN + 16) + (size_t)16 * KC * (T + 16) + (size_t)KC * (2 * TBH + 2) * 24);      // raw: [KC][RH][24]
}

// returns MRDIS_EUNSUPPORTED when the layer is outside what this kernel covers (caller falls back to the direct kernel)
int mrdis_run_wino(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                   int N, int H, int W, int Ci, int Co, int flip, int lrelu, hipStream_t s) {
    if (Ci % 4 != 0 || ldx % 4 != 0 || (((uintptr_t)x) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    WinoParams p{};
    p.in = x; p.w = w; p.bias = bias; p.out = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldy;
    p.flip = flip; p.lrelu = lrelu;
    { const long long mb = mrdis_opt(MRDIS_OPT_NT_MB); p.nt_out = (long long)N * H * W * ldy * 4 >= mb * 1000000LL ? 1 : 0; }
    const int CG = Co > 32 ? 2 : 1;
    p.TBH = 8; p.TBW = 8;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    p.nby = mrdis_cdiv(th, p.TBH); p.nbx = mrdis_cdiv(tw, p.TBW);
    p.coTiles = mrdis_cdiv(Co, 32 * CG);
    const long long nblk = (long long)N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wino_conv_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino_conv_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    const size_t lds = CG == 2 ? wino_lds<2, 4>(p.TBH, p.TBW) : wino_lds<1, 4>(p.TBH, p.TBW);
    mrdis_count(MRDIS_CNT_WINO);
    if (CG == 2) MRDIS_LAUNCH((wino_conv_kernel<2, 4>), dim3((int)nblk), dim3(512), lds, s, p);
    else MRDIS_LAUNCH((wino_conv_kernel<1, 4>), dim3((int)nblk), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// hybrid 3-D entry (see wino_conv_kernel D3): x (N, D, H, W, Ci) -> y (N, D, H, W, Co), 27-tap filter [27][Ci][Co]
int mrdis_run_wino3d(const float* x, int ldx, const float* w, const float* bias, const float* res, int ldres, float* y, int ldy,
                     int N, int D, int H, int W, int Ci, int Co, int flip, hipStream_t s) {
    if (Ci % 4 != 0 || ldx % 4 != 0 || (((uintptr_t)x) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * D * H * W * ldx >= 0x7fffffffLL || (long long)N * D * H * W * ldy >= 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    WinoParams p{};
    p.in = x; p.w = w; p.bias = bias; p.out = y; p.res = res; p.ldres = ldres;
    p.D = D; p.N = N * D; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldy;
    p.flip = flip; p.lrelu = 0;
    { const long long mb = mrdis_opt(MRDIS_OPT_NT_MB); p.nt_out = (long long)N * D * H * W * ldy * 4 >= mb * 1000000LL ? 1 : 0; }
    const int CG = Co > 32 ? 2 : 1;
    p.TBH = 8; p.TBW = 8;
    p.nby = mrdis_cdiv((H + 1) / 2, p.TBH); p.nbx = mrdis_cdiv((W + 1) / 2, p.TBW);
    p.coTiles = mrdis_cdiv(Co, 32 * CG);
    const long long nblk = (long long)p.N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wino_conv_kernel<1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino_conv_kernel<2, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    const size_t lds = CG == 2 ? wino_lds<2, 4>(p.TBH, p.TBW) : wino_lds<1, 4>(p.TBH, p.TBW);
    mrdis_count(MRDIS_CNT_WINO_SPADE);
    if (CG == 2) MRDIS_LAUNCH((wino_conv_kernel<2, 4, true>), dim3((int)nblk), dim3(512), lds, s, p);
    else MRDIS_LAUNCH((wino_conv_kernel<1, 4, true>), dim3((int)nblk), dim3(256), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// =========================================================================== Winograd weight gradient
// dW of a 3x3 / stride 1 / pad 1 layer through the same F(2x2, 3x3) factorisation: with M = U .* V and Y = A^T M A,
//
//      dU_xi[ci][co] = sum over tiles  V_xi[ci][tile] * Z_xi[tile][co],   Z = A dY A^T  (4x4 from the 2x2 dy tile)
//      dg = G^T dU G                                                      (3x3 from 4x4)
//
// i.e. 16 small GEMMs whose reduction axis is the TILE index -- 16/36 of the direct kernel's multiplies.  One kernel:
// a workgroup owns a (CIB x COB) block of (ci, co) and a grid-stride share of the tile blocks (split-K); per block of
// 8 tiles (2x4) the halo'd x block goes to LDS, every thread transforms one (tile, ci) to V and one (tile, co) -- dy read
// straight from global, prefetched -- to Z, then v_mfma_f32_16x16x4_f32 with A = V_xi[16 ci x 4 tiles],
// B = Z_xi[4 tiles x 16 co]; a wave owns 16 ci x 32 co for all 16 xi (128 accumulators), so G^T dU G is lane-local
// and a workgroup writes nine [ci][co] planes (not sixteen) into its split-K slab; a fixed-order sum over the slabs
// finishes dw_tck (bit-reproducible).  The bias gradient (column sums of dy) rides along in the Z pass.
// D3: slice kd of a 3x3x3 filter gradient = the 2-D gradient between dy plane d and x plane d + kd - 1 (three launches)
template <int WCI, int WCO, bool D3 = false>       // waves along ci (16 each) x waves along co (32 each)
__global__ __launch_bounds__(64 * WCI * WCO, (WCI * WCO == 8) ? 1 : 2) void wino_wgrad_kernel(const WinoWgradParams p) {
    constexpr int NT = 64 * WCI * WCO, CIB = 16 * WCI, COB = 32 * WCO, TB = 8, TBW = 4;
    constexpr int RH = 6, RW = 10, NPX = RH * RW;
    constexpr int RP = CIB + 4, VP = CIB + 16, ZP = COB + 16;
    constexpr int XQ = CIB / 4;                                   // float4 per raw pixel
    constexpr int XR = (NPX * XQ + NT - 1) / NT;
    constexpr int VR = (TB * CIB) / NT > 0 ? (TB * CIB) / NT : 1; // V items per thread
    constexpr int ZR = (TB * COB) / NT > 0 ? (TB * COB) / NT : 1; // Z items per thread
    static_assert((TB * CIB) % NT == 0 || TB * CIB < NT, "V items");
    static_assert((TB * COB) % NT == 0 || TB * COB < NT, "Z items");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* V = smem;                                              // [16][TB][VP]
    float* Z = V + 16 * TB * VP;                                  // [16][TB][ZP]
    float* raw = Z + 16 * TB * ZP;                                // [NPX][RP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int wi = wave % WCI, wo = wave / WCI;
    int b_ = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // the (cib, cob) workgroups of a split read the same x / dy tiles: same XCD, same L2
    const int cob = b_ % p.nCoB; b_ /= p.nCoB;
    const int cib = b_ % p.nCiB;
    const int split = b_ / p.nCiB;
    const int ci0 = cib * CIB, co0 = cob * COB;

    // block-invariant roles
    int xoff[XR];                                                 // (ry << 16) | (rx << 8) | q, or -1
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * NT;
        xoff[it] = -1;
        if (idx < NPX * XQ) { const int pi = idx / XQ, q = idx - pi * XQ; xoff[it] = ((pi / RW) << 16) | ((pi % RW) << 8) | q; }
    }
    float4 xr[XR];
    float dr[ZR][4];
    auto load_block = [&](int blk) {
        int t = blk;
        const int bx = t % p.nbx; t /= p.nbx;
        const int by = t % p.nby;
        const int n = t / p.nby;
        const int oy0 = 4 * by, ox0 = 8 * bx;                     // output origin of the 2 x 4 tile block
        const bool plane_ok = !D3 || (unsigned)(n % p.D + p.kd - 1) < (unsigned)p.D;
        const float* xn = p.x + (long long)(D3 ? n + p.kd - 1 : n) * p.H * p.W * p.ldx + ci0;
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            xr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (plane_ok && xoff[it] >= 0) {
                const int h = oy0 - 1 + (xoff[it] >> 16), w_ = ox0 - 1 + ((xoff[it] >> 8) & 255), q = xoff[it] & 255;
                if ((unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                    xr[it] = *reinterpret_cast<const float4*>(xn + ((long long)h * p.W + w_) * p.ldx + 4 * q);
            }
        }
        const float* dn = p.dy + (long long)n * p.H * p.W * p.lddy + co0;
#pragma unroll
        for (int z = 0; z < ZR; ++z) {
            const int item = tid + z * NT;
            const int co = item % COB, tile = item / COB;
            const int oy = oy0 + 2 * (tile / TBW), ox = ox0 + 2 * (tile % TBW);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    dr[z][2 * a + b] = (item < TB * COB && oy + a < p.H && ox + b < p.W) ? dn[((long long)(oy + a) * p.W + ox + b) * p.lddy + co] : 0.f;
        }
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float bsum[ZR];
#pragma unroll
    for (int z = 0; z < ZR; ++z) bsum[z] = 0.f;

    const int a_off = kq * VP + 16 * wi + l16;                    // + (xi * TB + 4 * ks) * VP
    const int b_off = kq * ZP + 32 * wo + l16;                    // + (xi * TB + 4 * ks) * ZP (+ 16 for the second block)

    int blk = split;
    if (blk < p.nblocks) load_block(blk);
    for (; blk < p.nblocks; blk += p.splits) {
        __syncthreads();                                          // the previous block's MFMAs have read V / Z; its V pass has read raw
        // raw x block -> LDS; Z = A dY A^T straight from the prefetched registers
#pragma unroll
        for (int it = 0; it < XR; ++it)
            if (xoff[it] >= 0) {
                const int pi = (xoff[it] >> 16) * RW + ((xoff[it] >> 8) & 255), q = xoff[it] & 255;
                *reinterpret_cast<float4*>(raw + pi * RP + 4 * q) = xr[it];
            }
#pragma unroll
        for (int z = 0; z < ZR; ++z) {
            const int item = tid + z * NT;
            if (item < TB * COB) {
                const int co = item % COB, tile = item / COB;
                const float d00 = dr[z][0], d01 = dr[z][1], d10 = dr[z][2], d11 = dr[z][3];
                bsum[z] += (d00 + d01) + (d10 + d11);
                const float t_[4][2] = {{d00, d01}, {d00 + d10, d01 + d11}, {d00 - d10, d01 - d11}, {-d10, -d11}};
                float* zp = Z + tile * ZP + co;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    zp[(4 * i + 0) * TB * ZP] = t_[i][0];
                    zp[(4 * i + 1) * TB * ZP] = t_[i][0] + t_[i][1];
                    zp[(4 * i + 2) * TB * ZP] = t_[i][0] - t_[i][1];
                    zp[(4 * i + 3) * TB * ZP] = -t_[i][1];
                }
            }
        }
        __syncthreads();
        // V = B^T d B per (tile, ci)
#pragma unroll
        for (int v = 0; v < VR; ++v) {
            const int item = tid + v * NT;
            if (item < TB * CIB) {
                const int ci = item % CIB, tile = item / CIB;
                const float* rp = raw + ((2 * (tile / TBW)) * RW + 2 * (tile % TBW)) * RP + ci;
                float d[4][4], r[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[i][j] = rp[(i * RW + j) * RP];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r[0][j] = d[0][j] - d[2][j]; r[1][j] = d[1][j] + d[2][j];
                    r[2][j] = d[2][j] - d[1][j]; r[3][j] = d[1][j] - d[3][j];
                }
                float* vp = V + tile * VP + ci;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    vp[(4 * a + 0) * TB * VP] = r[a][0] - r[a][2];
                    vp[(4 * a + 1) * TB * VP] = r[a][1] + r[a][2];
                    vp[(4 * a + 2) * TB * VP] = r[a][2] - r[a][1];
                    vp[(4 * a + 3) * TB * VP] = r[a][1] - r[a][3];
                }
            }
        }
        __syncthreads();
        if (blk + p.splits < p.nblocks) load_block(blk + p.splits);          // in flight during the MFMAs
        {
            constexpr int NS = 16 * (TB / 4);
            float av[3], b0[3], b1[3];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int row = (s_ / (TB / 4)) * TB + 4 * (s_ % (TB / 4));
                av[s_] = V[row * VP + a_off]; b0[s_] = Z[row * ZP + b_off]; b1[s_] = Z[row * ZP + b_off + 16];
            }
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                if (s_ + 2 < NS) {
                    const int row = ((s_ + 2) / (TB / 4)) * TB + 4 * ((s_ + 2) % (TB / 4));
                    av[(s_ + 2) % 3] = V[row * VP + a_off]; b0[(s_ + 2) % 3] = Z[row * ZP + b_off]; b1[(s_ + 2) % 3] = Z[row * ZP + b_off + 16];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int x = s_ / (TB / 4);
                acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_ % 3], b0[s_ % 3], acc[x][0], 0, 0, 0);
                acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_ % 3], b1[s_ % 3], acc[x][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: dg = G^T dU G per lane; D rows (4 kq + r) = ci, col l16 = co
    float* out = p.slab + (long long)split * 9 * p.Ci * p.Co;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int co = co0 + 32 * wo + 16 * b + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ci0 + 16 * wi + 4 * kq + r;
            // rows: G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
            float t_[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u0 = acc[j][b][r], u1 = acc[4 + j][b][r], u2 = acc[8 + j][b][r], u3 = acc[12 + j][b][r];
                t_[0][j] = u0 + 0.5f * (u1 + u2); t_[1][j] = 0.5f * (u1 - u2); t_[2][j] = 0.5f * (u1 + u2) + u3;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float g0 = t_[a][0] + 0.5f * (t_[a][1] + t_[a][2]);
                const float g1 = 0.5f * (t_[a][1] - t_[a][2]);
                const float g2 = 0.5f * (t_[a][1] + t_[a][2]) + t_[a][3];
                out[((long long)(3 * a + 0) * p.Ci + ci) * p.Co + co] = g0;
                out[((long long)(3 * a + 1) * p.Ci + ci) * p.Co + co] = g1;
                out[((long long)(3 * a + 2) * p.Ci + ci) * p.Co + co] = g2;
            }
        }
    }
    if (p.bias_slab != nullptr && cib == 0) {
        // items of a thread share the channel (NT % COB == 0): reduce over the threads with the same co through LDS
        __syncthreads();
        float s_ = 0.f;
#pragma unroll
        for (int z = 0; z < ZR; ++z) s_ += bsum[z];
        float* red = smem;                                        // [NT]
        red[tid] = (tid < TB * COB || ZR > 1) ? s_ : 0.f;
        __syncthreads();
        if (tid < COB) {
            float t = 0.f;
            for (int k = tid; k < NT; k += COB) t += red[k];
            p.bias_slab[(long long)split * p.Co + co0 + tid] = t;
        }
    }
}

// dw[i] = sum over the split-K slabs, fixed order: block (64, SL) -- lane y sums slabs y, y + SL, ... with four loads in
// flight, the SL partial sums meet in LDS in lane order (the first version walked all slabs in one thread: 0.3 TB/s)
__global__ void wino_sum_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int nslab,
                                      const float* __restrict__ bslab, float* __restrict__ dbias, int Co, int accumulate_bias) {
    __shared__ float red[16][65];
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    const int SL = blockDim.y, y = threadIdx.y;
    float s_ = 0.f;
    if (i < n) {
        const float* src = slab + i + (long long)y * n;
        const long long step = (long long)SL * n;
        int k = y;
        for (; k + 3 * SL < nslab; k += 4 * SL, src += 4 * step) { const float a = src[0], b = src[step], c = src[2 * step], d = src[3 * step]; s_ += a; s_ += b; s_ += c; s_ += d; }
        for (; k < nslab; k += SL, src += step) s_ += *src;
    } else if (dbias != nullptr && i < n + Co) {
        const int co = (int)(i - n);
        for (int k = y; k < nslab; k += SL) s_ += bslab[(long long)k * Co + co];
    }
    red[y][threadIdx.x] = s_;
    __syncthreads();
    if (y == 0) {
        float t = 0.f;
        for (int k = 0; k < SL; ++k) t += red[k][threadIdx.x];
        if (i < n) dw[i] = t;
        else if (dbias != nullptr && i < n + Co) { const int co = (int)(i - n); dbias[co] = accumulate_bias ? dbias[co] + t : t; }
    }
}

struct WinoWgradPlan { WinoWgradParams p; int wci, wco; size_t lds; bool ok; };

static void plan_wino_wgrad(WinoWgradPlan& pl, int N, int H, int W, int Ci, int Co) {
    pl.ok = false;
    int wci, wco;
    if (Ci % 64 == 0 && Co % 64 == 0) { wci = 4; wco = 2; }
    else if (Ci % 64 == 0 && Co % 32 == 0) { wci = 4; wco = 1; }
    else if (Ci % 32 == 0 && Co % 64 == 0) { wci = 2; wco = 2; }
    else return;
    WinoWgradParams& p = pl.p;
    p = WinoWgradParams{};
    p.N = N; p.H = H; p.W = W; p.Ci = Ci; p.Co = Co;
    p.nby = mrdis_cdiv((H + 1) / 2, 2); p.nbx = mrdis_cdiv((W + 1) / 2, 4);
    const long long nb = (long long)N * p.nby * p.nbx;
    if (nb > 0x7fffffffLL) return;
    p.nblocks = (int)nb;
    const int CIB = 16 * wci, COB = 32 * wco;
    p.nCiB = Ci / CIB; p.nCoB = Co / COB;
    const int target = (wci * wco == 8) ? 256 : 512;
    int splits = target / (p.nCiB * p.nCoB);
    if (splits < 1) splits = 1;
    if (splits > p.nblocks) splits = p.nblocks;
    p.splits = splits;
    pl.wci = wci; pl.wco = wco;
    pl.lds = sizeof(float) * ((size_t)16 * 8 * (CIB + 16) + (size_t)16 * 8 * (COB + 16) + (size_t)60 * (CIB + 4));
    pl.ok = true;
}

size_t mrdis_wino_wgrad_workspace(int N, int H, int W, int Ci, int Co) {
    WinoWgradPlan pl;
    plan_wino_wgrad(pl, N, H, W, Ci, Co);
    if (!pl.ok) return 0;
    return sizeof(float) * ((size_t)pl.p.splits * 9 * Ci * Co + (size_t)pl.p.splits * Co) + 256;
}

// returns MRDIS_EUNSUPPORTED when the layer is outside what this kernel covers
int mrdis_run_wino_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                         size_t workspace_bytes, int N, int H, int W, int Ci, int Co, int accumulate_bias, hipStream_t s) {
    WinoWgradPlan pl;
    plan_wino_wgrad(pl, N, H, W, Ci, Co);
    if (!pl.ok || ldx % 4 != 0 || (((uintptr_t)x) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes < mrdis_wino_wgrad_workspace(N, H, W, Ci, Co) - 256) return MRDIS_EUNSUPPORTED;
    WinoWgradParams& p = pl.p;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy;
    p.slab = reinterpret_cast<float*>(workspace);
    p.bias_slab = dbias ? p.slab + (size_t)p.splits * 9 * Ci * Co : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wino_wgrad_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino_wgrad_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino_wgrad_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    int rc2 = MRDIS_EUNSUPPORTED;
    {   // Winograd F(3x3, 4x4) (mrdis_wino4w.hip): 1.78x fewer MFMAs; writes its own number of slabs (<= the plan's) and sets p.splits to it
        const int rc4 = mrdis_launch_wino4_wgrad(p, p.splits, s);
        if (rc4 == MRDIS_OK) rc2 = MRDIS_OK;
        else if (rc4 != MRDIS_EUNSUPPORTED) return rc4;
    }
    const int nblk = p.splits * p.nCiB * p.nCoB;
    if (rc2 != MRDIS_OK && pl.wci == 4 && pl.wco == 2 && mrdis_opt(MRDIS_OPT_WINO_PIPE)) {         // software-pipelined form (mrdis_wino2.hip)
        rc2 = mrdis_launch_wino_wgrad2(p, s);
        if (rc2 != MRDIS_OK && rc2 != MRDIS_EUNSUPPORTED) return rc2;
    }
    if (rc2 == MRDIS_OK) {}
    else if (pl.wci == 4 && pl.wco == 2) { mrdis_count(MRDIS_CNT_WINO_WGRAD); MRDIS_LAUNCH((wino_wgrad_kernel<4, 2>), dim3(nblk), dim3(512), pl.lds, s, p); }
    else if (pl.wci == 4) { mrdis_count(MRDIS_CNT_WINO_WGRAD); MRDIS_LAUNCH((wino_wgrad_kernel<4, 1>), dim3(nblk), dim3(256), pl.lds, s, p); }
    else { mrdis_count(MRDIS_CNT_WINO_WGRAD); MRDIS_LAUNCH((wino_wgrad_kernel<2, 2>), dim3(nblk), dim3(256), pl.lds, s, p); }
    MRDIS_CHECK_LAUNCH();
    const long long n = 9LL * Ci * Co;
    int SL = 1;
    while (SL < 16 && SL * 4 <= p.splits) SL <<= 1;
    MRDIS_LAUNCH(wino_sum_slabs_kernel, dim3(mrdis_cdiv(n + (dbias ? Co : 0), 64)), dim3(64, SL), 0, s, p.slab, dw_tck, n, p.splits,
                       p.bias_slab, dbias, Co, accumulate_bias);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// hybrid 3-D weight gradient: dw_tck [27][Ci][Co], three launches of the 2-D kernel (one per depth tap)
size_t mrdis_wino_wgrad3d_workspace(int N, int D, int H, int W, int Ci, int Co) { return mrdis_wino_wgrad_workspace(N * D, H, W, Ci, Co); }

int mrdis_run_wino_wgrad3d(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace,
                           size_t workspace_bytes, int N, int D, int H, int W, int Ci, int Co, hipStream_t s) {
    WinoWgradPlan pl;
    plan_wino_wgrad(pl, N * D, H, W, Ci, Co);
    if (!pl.ok || ldx % 4 != 0 || (((uintptr_t)x) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_wino_wgrad3d_workspace(N, D, H, W, Ci, Co)) return MRDIS_EUNSUPPORTED;
    WinoWgradParams& p = pl.p;
    p.x = x; p.dy = dy; p.ldx = ldx; p.lddy = lddy; p.D = D;
    p.slab = reinterpret_cast<float*>(workspace);
    float* bias_slab = p.slab + (size_t)p.splits * 9 * Ci * Co;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wino_wgrad_kernel<4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino_wgrad_kernel<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino_wgrad_kernel<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return MRDIS_ELAUNCH;
        attr_set = true;
    }
    const int nblk = p.splits * p.nCiB * p.nCoB;
    const long long n = 9LL * Ci * Co;
    int SL = 1;
    while (SL < 16 && SL * 4 <= p.splits) SL <<= 1;
    for (int kd = 0; kd < 3; ++kd) {
        p.kd = kd;
        p.bias_slab = (dbias && kd == 1) ? bias_slab : nullptr;
        if (pl.wci == 4 && pl.wco == 2) MRDIS_LAUNCH((wino_wgrad_kernel<4, 2, true>), dim3(nblk), dim3(512), pl.lds, s, p);
        else if (pl.wci == 4) MRDIS_LAUNCH((wino_wgrad_kernel<4, 1, true>), dim3(nblk), dim3(256), pl.lds, s, p);
        else MRDIS_LAUNCH((wino_wgrad_kernel<2, 2, true>), dim3(nblk), dim3(256), pl.lds, s, p);
        MRDIS_CHECK_LAUNCH();
        float* db = p.bias_slab ? dbias : nullptr;
        MRDIS_LAUNCH(wino_sum_slabs_kernel, dim3(mrdis_cdiv(n + (db ? Co : 0), 64)), dim3(64, SL), 0, s, p.slab, dw_tck + kd * n, n, p.splits,
                           p.bias_slab, db, Co, 0);
        MRDIS_CHECK_LAUNCH();
    }
    return MRDIS_OK;
}
