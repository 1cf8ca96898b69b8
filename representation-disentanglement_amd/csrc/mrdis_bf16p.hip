// mrdis_bf16p.hip -- software-pipelined bf16 convolution for the 3x3 / stride 1 / pad 1 layers with bf16 activations
// (MRDIS_DT_BF16), Cin a multiple of 32, maps at least 32 wide: the layers that are most of `compute_dtype: bf16`.
//
// bconv_kernel (mrdis_bf16.hip) runs an item (256 positions x 64 couts x 32 channels x 9 taps) as: barrier | registers -> LDS |
// barrier | issue next item's loads | 72 MFMAs.  Timing-only builds of it on 128 -> 256 at 64x64, B = 32 (tools/bconv_abl.py):
// 115 us = ~28 us of MFMAs + operand reads, ~35 us waiting on the global loads (fifteen 16-byte loads per thread issued in one
// burst), ~10 us of LDS stores, ~40 us of loop skeleton (address arithmetic of the staging, barriers, epilogue) -- back to back, although
// two workgroups share a CU.  Here the pipeline of wino2_kernel (mrdis_wino2.hip) carries the same arithmetic:
//
//   * ONE persistent workgroup of 8 waves per CU walks (tile, cout tile) units and their 32-channel chunks; a wave owns one row of
//     32 positions of the 8 x 32 tile x all 32 WC couts (v_mfma_f32_32x32x16_bf16, A = filter, B = pixels: D[cout][position]);
//   * the LDS images xs[pixel][32 + 8] and ws[tap][cout][32 + 8] (bf16) are double-buffered and there is ONE barrier per item:
//     while item i is multiplied, item i+1 goes from registers to LDS and the loads of item i+2 are issued, one 16-byte load or
//     store per (tap, k-step) of the MFMA sequence;
//   * every global load is a buffer load whose offset lies beyond the record count where there is nothing to read (zero padding,
//     ragged tiles, cout tails): no branch, counted vmcnt waits; staging offsets are item-invariant per thread plus two scalars;
//   * the bias sits in LDS; the epilogue converts to bf16 and writes 8 bytes per lane and 4-cout group.
//
// Results are bit-identical to bconv_kernel's (same products, same accumulation order).
#include "mrdis_tapconv.h"

typedef __bf16 bp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bp_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned bp_u32x4 __attribute__((ext_vector_type(4)));

typedef unsigned bp_u32x2 __attribute__((ext_vector_type(2)));
namespace {
// The two half-waves of a lane pair (e, half) hold the cout groups 8 q + 4 half .. + 3 of one position: two 8-byte pieces per 8 couts.  Swapping
// group q of the upper half with group q + 1 of the lower half (v_permlane32_swap, one instruction per dword) leaves every lane with 8
// CONSECUTIVE couts -- half 0: 8 q .. 8 q + 7, half 1: 8 (q + 1) .. 8 (q + 1) + 7 -- i.e. one 16-byte store per lane where there were two 8-byte
// stores: half as many write requests of twice the size reach L2 (the epilogue is bound by them: 80 of 140 us on the 256x256 level).
__device__ __forceinline__ bp_u32x4 bp_pair8(bp_u32x2 gq, bp_u32x2 gq1) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(gq[0], gq1[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(gq[1], gq1[1], false, false);
    return bp_u32x4{s0[0], s1[0], s0[1], s1[1]};
}
constexpr int P_KC = 32, P_PITCH = P_KC + 8, P_TH = 8, P_TW = 32, P_TINH = P_TH + 2, P_TINW = P_TW + 2, P_NPIX = P_TINH * P_TINW;
constexpr int P_XS = P_NPIX * P_PITCH;                 // bf16 elements of one x image
constexpr int P_BIAS = 1024;
constexpr unsigned P_OOB = 0xfffffff0u;
template <int V_> struct PIC { static constexpr int value = V_; };
__device__ __forceinline__ int p_opaque(int idx) { asm volatile("" : "+v"(idx)); return idx; }
}  // namespace

struct BConv3Params {
    const void* in; const void* w; const float* bias; void* out;       // bf16 activations, bf16 filter [tap][Cout][Cin], fp32 bias
    int N, H, W, Cin, ldin, Cout, ldout;
    int tilesA, tilesB, coTiles, units;               // units = N * tilesA * tilesB * coTiles
    int nchunks, lrelu;
    int dh[9], dw[9], widx[9];
    unsigned in_bytes, w_bytes;
    // SPADE epilogue (bconv3_kernel<2, ., true>): the filter is a fused gamma | beta filter (Cout = 2 C); a workgroup owns 32 channels and their
    // 32 gamma + 32 beta couts (accumulator block 0 / 1 of a lane: gamma / beta of the same channels); the epilogue writes
    // (z - mean) * rstd * (1 + gamma) + beta and gamma (model.py:2440-2446)
    const void* z; const float* mean; const float* rstd; void* gamma_out;
    int ldz, ldg, C;
    int wide;                                         // 16-byte output stores: Cout (C) % 8 == 0, output views 16-byte aligned with ld % 8 == 0
};

// ABL (timing-only, -DBCONV3_ABLATIONS): 1 no MFMAs, 2 no global loads, 4 no LDS stores, 8 operand reads at one address, 16 no output stores
template <int WC, int ABL = 0, bool SPADE = false>
__global__ __launch_bounds__(512, 1) void bconv3_kernel(const BConv3Params p) {
    static_assert(!SPADE || WC == 2, "SPADE: 32 gamma + 32 beta couts per workgroup");
    constexpr int BN = 32 * WC, NT = 512;
    constexpr int WS = 9 * BN * P_PITCH;              // bf16 elements of one filter image
    constexpr int XR = (P_NPIX * 4 + NT - 1) / NT;    // 16-byte x pieces per thread (3)
    constexpr int WR = (9 * BN * 4 + NT - 1) / NT;    // 16-byte filter pieces per thread (5 | 3)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* const lds = reinterpret_cast<__bf16*>(smem_raw);      // [2][WS] filter images, then [2][P_XS] x images, then the bias (fp32)
    float* const Bs = reinterpret_cast<float*>(lds + 2 * WS + 2 * P_XS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, e = lane & 31;
    // MFMA role: positions (ty = wave, tx = e) of the 8 x 32 tile, k-group `half`
    const int b_base = (wave * P_TINW + e) * P_PITCH + 8 * half;          // + tap offset + 16 ks, in the x image
    const int a_base = e * P_PITCH + 8 * half;                            // + (tap * BN + 32 j) * PITCH + 16 ks, in the filter image
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) toff[t] = ((p.dh[t] + 1) * P_TINW + (p.dw[t] + 1)) * P_PITCH;

    // staging roles (item-invariant): LDS element offset of each piece and its global offset relative to the item's origin
    int x_lds[XR]; unsigned x_rel[XR]; int x_yx[XR];   // x_yx = (iy << 8) | ix, or -1
#pragma unroll
    for (int it = 0; it < XR; ++it) {
        const int idx = tid + it * NT, pi = idx >> 2, q = idx & 3;
        const int iy = pi / P_TINW, ix = pi - iy * P_TINW;
        x_yx[it] = (idx < P_NPIX * 4) ? ((iy << 8) | ix) : -1;
        x_lds[it] = pi * P_PITCH + 8 * q;
        x_rel[it] = 2u * (unsigned)((iy * p.W + ix) * p.ldin + 8 * q);
    }
    int w_lds[WR]; unsigned w_rel[WR]; int w_co[WR];   // w_co: cout within the tile, or -1
#pragma unroll
    for (int it = 0; it < WR; ++it) {
        const int idx = tid + it * NT, row = idx >> 2, q = idx & 3;
        const int t = row / BN, co = row - t * BN;
        w_co[it] = (idx < 9 * BN * 4) ? co : -1;
        w_lds[it] = row * P_PITCH + 8 * q;
        // SPADE: local couts 0..31 are the gamma couts of the workgroup's 32 channels, 32..63 their beta couts (C further on in the filter)
        const int co_g = SPADE ? ((co >> 5) ? p.C : 0) + (co & 31) : co;
        w_rel[it] = 2u * (unsigned)((p.widx[t < 9 ? t : 0] * p.Cout + co_g) * p.Cin + 8 * q);
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    for (int c = tid; c < P_BIAS; c += NT) Bs[c] = (p.bias != nullptr && c < p.Cout) ? p.bias[c] : 0.f;

    const int grid = gridDim.x;
    const int u0 = mrdis_xcd_remap(blockIdx.x, grid);
    const int nmine = (p.units - u0 + grid - 1) / grid;           // host: grid <= units
    const int total = nmine * p.nchunks;
    auto decode = [&](int j, int& n, int& a0, int& b0, int& co0) {
        int u = u0 + j * grid;
        const int cot = u % p.coTiles; u /= p.coTiles;
        const int tb = u % p.tilesB; u /= p.tilesB;
        const int ta = u % p.tilesA;
        n = u / p.tilesA; a0 = ta * P_TH; b0 = tb * P_TW; co0 = cot * BN;
    };

    // ---- load cursor: unit lj, chunk lc; per-unit scalars recomputed when the unit changes
    int lj = 0, lc = 0;
    int l_h0 = 0, l_w0 = 0, l_co0 = 0; unsigned l_xorg = 0; bool l_live = false;
    auto load_unit = [&]() {
        l_live = lj < nmine;
        if (l_live) {
            int n, a0, b0, co0; decode(lj, n, a0, b0, co0);
            l_h0 = a0 - 1; l_w0 = b0 - 1; l_co0 = co0;
            l_xorg = 2u * (unsigned)(((n * p.H + l_h0) * p.W + l_w0) * p.ldin);        // wraps for halo origins; added mod 2^32 below
        }
    };
    bp_u32x4 xr[2][XR], wr[2][WR];
    unsigned xo[XR], wo[WR];                           // byte offsets of the NEXT item's pieces
    auto next_offsets = [&]() {
        const unsigned c0b = 2u * (unsigned)(lc * P_KC);
#pragma unroll
        for (int it = 0; it < XR; ++it) {
            const int h = l_h0 + (x_yx[it] >> 8), w_ = l_w0 + (x_yx[it] & 255);
            const bool ok = l_live && x_yx[it] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
            xo[it] = ok ? l_xorg + x_rel[it] + c0b : P_OOB;
        }
#pragma unroll
        for (int it = 0; it < WR; ++it) {
            const bool ok = l_live && w_co[it] >= 0 && (SPADE ? l_co0 / 2 + (w_co[it] & 31) < p.C : l_co0 + w_co[it] < p.Cout);
            wo[it] = ok ? w_rel[it] + 2u * (unsigned)((SPADE ? l_co0 / 2 : l_co0) * p.Cin) + c0b : P_OOB;
        }
        if (++lc == p.nchunks) { lc = 0; ++lj; load_unit(); }
    };
    auto load_x = [&](auto S_, int it) { constexpr int S = decltype(S_)::value; if (ABL & 2) xr[S][it] = bp_u32x4{0u, 0u, 0u, 0u}; else xr[S][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)xo[it], 0, 0); };
    auto load_w = [&](auto S_, int it) { constexpr int S = decltype(S_)::value; if (ABL & 2) wr[S][it] = bp_u32x4{0u, 0u, 0u, 0u}; else wr[S][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)wo[it], 0, 0); };
    auto store_x = [&](auto S_, __bf16* xs, int it) {
        constexpr int S = decltype(S_)::value;
        if (x_yx[it] >= 0 && !(ABL & 4)) *reinterpret_cast<bp_u32x4*>(xs + x_lds[it]) = xr[S][it];
    };
    auto store_w = [&](auto S_, __bf16* ws, int it) {
        constexpr int S = decltype(S_)::value;
        if (w_co[it] >= 0 && !(ABL & 4)) *reinterpret_cast<bp_u32x4*>(ws + w_lds[it]) = wr[S][it];
    };

    f32x16 acc[WC];
#pragma unroll
    for (int j = 0; j < WC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- prologue: item 0 in LDS buffer 0, item 1 in register set 1, cursor at item 2
    load_unit();
    next_offsets();
#pragma unroll
    for (int it = 0; it < XR; ++it) load_x(PIC<0>{}, it);
#pragma unroll
    for (int it = 0; it < WR; ++it) load_w(PIC<0>{}, it);
    next_offsets();
#pragma unroll
    for (int it = 0; it < XR; ++it) load_x(PIC<1>{}, it);
#pragma unroll
    for (int it = 0; it < WR; ++it) load_w(PIC<1>{}, it);
#pragma unroll
    for (int it = 0; it < XR; ++it) store_x(PIC<0>{}, lds + 2 * WS, it);
#pragma unroll
    for (int it = 0; it < WR; ++it) store_w(PIC<0>{}, lds, it);
    __syncthreads();

    int mj = 0, mc = 0;
    auto iteration = [&](auto P_) {
        constexpr int P = decltype(P_)::value;
        const __bf16* wa = lds + p_opaque(P * WS + a_base);
        const __bf16* xb = lds + p_opaque(2 * WS + P * P_XS + b_base);
        __bf16* wn = lds + (P ^ 1) * WS;              // item i + 1 goes here (from register set P ^ 1)
        __bf16* xn = lds + 2 * WS + (P ^ 1) * P_XS;
        next_offsets();                               // item i + 2: loaded into register set P in the first XR + WR steps
        bp_bf16x8 af[2][WC], bf[2];
#pragma unroll
        for (int j = 0; j < WC; ++j) af[0][j] = *reinterpret_cast<const bp_bf16x8*>(wa + 32 * j * P_PITCH);
        bf[0] = *reinterpret_cast<const bp_bf16x8*>(xb + toff[0]);
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            const int t = s_ >> 1, ks = s_ & 1, c_ = s_ & 1;
            if (s_ + 1 < 18 && !(ABL & 8)) {
                const int t1 = (s_ + 1) >> 1, ks1 = (s_ + 1) & 1;
#pragma unroll
                for (int j = 0; j < WC; ++j) af[c_ ^ 1][j] = *reinterpret_cast<const bp_bf16x8*>(wa + (t1 * BN + 32 * j) * P_PITCH + 16 * ks1);
                bf[c_ ^ 1] = *reinterpret_cast<const bp_bf16x8*>(xb + toff[t1] + 16 * ks1);
            }
            // staging slice of this step: loads of item i + 2 first, then the LDS stores of item i + 1
            if (s_ < XR) load_x(PIC<P>{}, s_);
            else if (s_ < XR + WR) load_w(PIC<P>{}, s_ - XR);
            else if (s_ < 2 * XR + WR) store_x(PIC<P ^ 1>{}, xn, s_ - XR - WR);
            else if (s_ < 2 * XR + 2 * WR) store_w(PIC<P ^ 1>{}, wn, s_ - 2 * XR - WR);
            (void)t; (void)ks;
#pragma unroll
            for (int j = 0; j < WC; ++j) { if (ABL & 1) acc[j][0] += (float)af[(ABL & 8) ? 0 : c_][j][0] * (float)bf[(ABL & 8) ? 0 : c_][0]; else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(ABL & 8) ? 0 : c_][j], bf[(ABL & 8) ? 0 : c_], acc[j], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();

        if (++mc == p.nchunks) {
            // ---- epilogue of unit mj: D[cout][position]; a lane owns position (wave, e) and the couts 8 g + 4 half .. + 3 (g = 0..3) of each block
            int n, a0, b0, co0; decode(mj, n, a0, b0, co0);
            mc = 0; ++mj;
            const int a = a0 + wave, b = b0 + e;
            const bool pos_ok = a < p.H && b < p.W;
            if (SPADE) {
                const int c0 = co0 / 2;                          // the workgroup's first channel
                const long long pix = (long long)(n * p.H + a) * p.W + b;
                const __bf16* zp = reinterpret_cast<const __bf16*>(p.z) + pix * p.ldz;
                bp_bf16x4 zq[4]; float4 mu[4], rs[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {                  // loads first
                    const int ch = c0 + 8 * q + 4 * half;
                    const bool ok = pos_ok && ch < p.C;
                    zq[q] = ok ? *reinterpret_cast<const bp_bf16x4*>(zp + ch) : bp_bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    mu[q] = ch < p.C ? *reinterpret_cast<const float4*>(p.mean + (long long)n * p.C + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    rs[q] = ch < p.C ? *reinterpret_cast<const float4*>(p.rstd + (long long)n * p.C + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                __bf16* mixp = reinterpret_cast<__bf16*>(p.out) + pix * p.ldout;
                __bf16* gamp = reinterpret_cast<__bf16*>(p.gamma_out) + pix * p.ldg;
                bp_u32x2 pko[4], pkg[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = c0 + 8 * q + 4 * half;
                    const float4 bg = *reinterpret_cast<const float4*>(Bs + (ch < p.C ? ch : 0));
                    const float4 bb = *reinterpret_cast<const float4*>(Bs + (ch < p.C ? p.C + ch : 0));
                    const float g[4] = {acc[0][4 * q] + bg.x, acc[0][4 * q + 1] + bg.y, acc[0][4 * q + 2] + bg.z, acc[0][4 * q + 3] + bg.w};
                    const float bt[4] = {acc[WC - 1][4 * q] + bb.x, acc[WC - 1][4 * q + 1] + bb.y, acc[WC - 1][4 * q + 2] + bb.z, acc[WC - 1][4 * q + 3] + bb.w};
                    const float m_[4] = {mu[q].x, mu[q].y, mu[q].z, mu[q].w}, r_[4] = {rs[q].x, rs[q].y, rs[q].z, rs[q].w};
                    bp_bf16x4 o, og;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // gamma and beta reach the modulation kernel of the two-step path as bf16: round them the same way
                        const float gr = (float)(__bf16)g[k], br = (float)(__bf16)bt[k];
                        og[k] = (__bf16)g[k];
                        o[k] = (__bf16)(((float)zq[q][k] - m_[k]) * r_[k] * (1.f + gr) + br);
                    }
                    if (p.wide) { pko[q] = __builtin_bit_cast(bp_u32x2, o); pkg[q] = __builtin_bit_cast(bp_u32x2, og); }
                    else if (pos_ok && ch < p.C) { *reinterpret_cast<bp_bf16x4*>(mixp + ch) = o; *reinterpret_cast<bp_bf16x4*>(gamp + ch) = og; }
                }
                if (p.wide) {
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const bp_u32x4 wo = bp_pair8(pko[q], pko[q + 1]), wg = bp_pair8(pkg[q], pkg[q + 1]);
                        const int ch = c0 + 8 * (q + half);
                        if (pos_ok && ch < p.C) { *reinterpret_cast<bp_u32x4*>(mixp + ch) = wo; *reinterpret_cast<bp_u32x4*>(gamp + ch) = wg; }
                    }
                }
#pragma unroll
                for (int j = 0; j < WC; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                return;
            }
            __bf16* dst = reinterpret_cast<__bf16*>(p.out) + ((long long)(n * p.H + a) * p.W + b) * p.ldout;
            if (p.wide && !(ABL & 16)) {
#pragma unroll
                for (int j = 0; j < WC; ++j) {
                    bp_u32x2 pk[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = co0 + 32 * j + 8 * q + 4 * half;
                        const float4 bb = *reinterpret_cast<const float4*>(Bs + (co < P_BIAS - 3 ? co : 0));
                        float v[4] = {acc[j][4 * q] + bb.x, acc[j][4 * q + 1] + bb.y, acc[j][4 * q + 2] + bb.z, acc[j][4 * q + 3] + bb.w};
                        if (p.lrelu) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.2f * v[k];
                        }
                        bp_bf16x4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
                        pk[q] = __builtin_bit_cast(bp_u32x2, o);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        const bp_u32x4 w8 = bp_pair8(pk[q], pk[q + 1]);
                        const int co = co0 + 32 * j + 8 * (q + half);          // this lane's eight consecutive couts
                        if (pos_ok && co < p.Cout) *reinterpret_cast<bp_u32x4*>(dst + co) = w8;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < WC; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = co0 + 32 * j + 8 * q + 4 * half;
                    const float4 bb = *reinterpret_cast<const float4*>(Bs + (co < P_BIAS - 3 ? co : 0));
                    float v[4] = {acc[j][4 * q] + bb.x, acc[j][4 * q + 1] + bb.y, acc[j][4 * q + 2] + bb.z, acc[j][4 * q + 3] + bb.w};
                    if (p.lrelu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.2f * v[k];
                    }
                    bp_bf16x4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
                    if (pos_ok && co < p.Cout && (!(ABL & 16) || v[0] == 1.2345f)) *reinterpret_cast<bp_bf16x4*>(dst + co) = o;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            }
        }
    };
    for (int i = 0; i < total; i += 2) {
        iteration(PIC<0>{});
        if (i + 1 < total) iteration(PIC<1>{});
    }
}

// MRDIS_EUNSUPPORTED: the caller (mrdis_run_bconv) takes bconv_kernel
int mrdis_run_bconv3(const TapConvParams& t, hipStream_t s) {
    if (t.dtype != MRDIS_DT_BF16 || !t.w_bf16 || t.ntaps != 9 || t.is != 1 || t.os != 1 || t.oh0 != 0 || t.ow0 != 0) return MRDIS_EUNSUPPORTED;
    if (t.A != t.Hin || t.B != t.Win || t.Hout != t.Hin || t.Wout != t.Win) return MRDIS_EUNSUPPORTED;
    if (t.Cin % 32 != 0 || t.Cout % 4 != 0 || t.Cout < 16 || t.Cout > P_BIAS || t.Win < 32 || t.ldin % 8 != 0 || t.ldout % 4 != 0) return MRDIS_EUNSUPPORTED;
    if ((((uintptr_t)t.in | (uintptr_t)t.w_bf16) & 15) != 0 || ((uintptr_t)t.out & 7) != 0) return MRDIS_EUNSUPPORTED;
    BConv3Params p{};
    int wt = 0;
    for (int k = 0; k < 9; ++k) {
        if (t.dh[k] < -1 || t.dh[k] > 1 || t.dw[k] < -1 || t.dw[k] > 1) return MRDIS_EUNSUPPORTED;
        p.dh[k] = t.dh[k]; p.dw[k] = t.dw[k]; p.widx[k] = t.widx[k];
        if (t.widx[k] + 1 > wt) wt = t.widx[k] + 1;
    }
    const long long in_b = 2LL * (((long long)t.N * t.Hin * t.Win - 1) * t.ldin + t.Cin), w_b = 2LL * wt * t.Cin * t.Cout;
    if (in_b >= 0xffffffe0LL || w_b >= 0xffffffe0LL) return MRDIS_EUNSUPPORTED;
    p.in = t.in; p.w = t.w_bf16; p.bias = t.bias; p.out = t.out;
    p.N = t.N; p.H = t.Hin; p.W = t.Win; p.Cin = t.Cin; p.ldin = t.ldin; p.Cout = t.Cout; p.ldout = t.ldout;
    p.in_bytes = (unsigned)in_b; p.w_bytes = (unsigned)w_b;
    const int WC = t.Cout > 32 ? 2 : 1, BN = 32 * WC;
    p.tilesA = mrdis_cdiv(t.Hin, P_TH); p.tilesB = mrdis_cdiv(t.Win, P_TW); p.coTiles = mrdis_cdiv(t.Cout, BN);
    const long long units = (long long)t.N * p.tilesA * p.tilesB * p.coTiles;
    if (units > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.units = (int)units; p.nchunks = t.Cin / P_KC; p.lrelu = (t.epilogue & MRDIS_EPI_LRELU) ? 1 : 0;
    p.wide = (t.Cout % 8 == 0 && t.ldout % 8 == 0 && ((uintptr_t)t.out & 15) == 0 && !mrdis_opt(MRDIS_OPT_NOPACK)) ? 1 : 0;      // (debug_nopack = 1: 8-byte stores, as before)
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)bconv3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MRDIS_EUNSUPPORTED;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = units < n_cu ? (int)units : n_cu;
    const size_t lds = 2 * (size_t)(2 * 9 * BN * P_PITCH + 2 * P_XS) + sizeof(float) * P_BIAS;
#ifdef BCONV3_ABLATIONS
    if (WC == 2) {
        const int abl = (int)mrdis_opt(MRDIS_OPT_MODE);
#define BA(a) if (abl == a) { (void)hipFuncSetAttribute((const void*)bconv3_kernel<2, a>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        MRDIS_LAUNCH((bconv3_kernel<2, a>), dim3(grid), dim3(512), lds, s, p); MRDIS_CHECK_LAUNCH(); return MRDIS_OK; }
        BA(1) BA(2) BA(4) BA(8) BA(16) BA(6) BA(14) BA(15) BA(30)
#undef BA
    }
#endif
    mrdis_count(MRDIS_CNT_BCONV3);
    if (WC == 2) MRDIS_LAUNCH(bconv3_kernel<2>, dim3(grid), dim3(512), lds, s, p);
    else MRDIS_LAUNCH(bconv3_kernel<1>, dim3(grid), dim3(512), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// SPADE-fused form on bf16 activations (see BConv3Params): x = si_out (N, H, W, Ci) bf16, w = bf16 fused gamma | beta filter [9][2C][Ci],
// bias (2 C) fp32, z (N, H, W, C) bf16 with its instance statistics; writes mix and gamma (bf16).  MRDIS_EUNSUPPORTED: two-step path.
int mrdis_run_bconv3_spade(const void* x, int ldx, const void* w_bf16, const float* bias, const void* z, int ldz, const float* mean, const float* rstd,
                           void* mix, int ldmix, void* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s) {
    if (!mrdis_opt(MRDIS_OPT_WINO_PIPE) || Ci % 32 != 0 || C % 4 != 0 || C < 16 || 2 * C > P_BIAS || W < 32) return MRDIS_EUNSUPPORTED;
    if (ldx % 8 != 0 || ldz % 4 != 0 || ldmix % 4 != 0 || ldg % 4 != 0) return MRDIS_EUNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)w_bf16) | ((uintptr_t)mean) | ((uintptr_t)rstd)) & 15) != 0 || ((((uintptr_t)z) | ((uintptr_t)mix) | ((uintptr_t)gamma)) & 7) != 0) return MRDIS_EUNSUPPORTED;
    BConv3Params p{};
    for (int k = 0; k < 9; ++k) { p.dh[k] = k / 3 - 1; p.dw[k] = k % 3 - 1; p.widx[k] = k; }
    const long long in_b = 2LL * (((long long)N * H * W - 1) * ldx + Ci), w_b = 2LL * 9 * Ci * 2 * C;
    if (in_b >= 0xffffffe0LL || w_b >= 0xffffffe0LL) return MRDIS_EUNSUPPORTED;
    p.in = x; p.w = w_bf16; p.bias = bias; p.out = mix;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = 2 * C; p.ldout = ldmix;
    p.z = z; p.ldz = ldz; p.mean = mean; p.rstd = rstd; p.gamma_out = gamma; p.ldg = ldg; p.C = C;
    p.in_bytes = (unsigned)in_b; p.w_bytes = (unsigned)w_b;
    p.tilesA = mrdis_cdiv(H, P_TH); p.tilesB = mrdis_cdiv(W, P_TW); p.coTiles = mrdis_cdiv(C, 32);
    const long long units = (long long)N * p.tilesA * p.tilesB * p.coTiles;
    if (units > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.units = (int)units; p.nchunks = Ci / P_KC; p.lrelu = 0;
    p.wide = (C % 8 == 0 && ldmix % 8 == 0 && ldg % 8 == 0 && ((((uintptr_t)mix) | ((uintptr_t)gamma)) & 15) == 0 && !mrdis_opt(MRDIS_OPT_NOPACK)) ? 1 : 0;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)bconv3_kernel<2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return MRDIS_EUNSUPPORTED;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = units < n_cu ? (int)units : n_cu;
    const size_t lds = 2 * (size_t)(2 * 9 * 64 * P_PITCH + 2 * P_XS) + sizeof(float) * P_BIAS;
    mrdis_count(MRDIS_CNT_BCONV3_SPADE);
    MRDIS_LAUNCH((bconv3_kernel<2, 0, true>), dim3(grid), dim3(512), lds, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
