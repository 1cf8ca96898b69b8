// mrdis_wino2.hip -- software-pipelined fused Winograd F(2x2, 3x3) convolution for gfx950 (MI355X), fp32, NHWC, Cout > 32.
//
// Same arithmetic as wino_conv_kernel<2, 4> (mrdis_wino.hip): a workgroup of 8 waves owns 64 tiles (8x8 tiles = 16x16 outputs)
// x 64 couts, a wave 16 tiles x 32 couts for all 16 Winograd points (128 accumulators), 8 input channels per chunk.  That
// kernel runs its phases back to back -- raw block + U to LDS | barrier | V transform | barrier | 64 MFMAs | barrier -- and
// its matrix pipe is busy 47-51 % of the time (profiles/r02_pmc_mfma.md).  Here the phases of DIFFERENT chunks overlap
// inside every wave:
//
//   iteration i (one barrier):   64 MFMAs on U(i), V(i)                                   <- matrix pipe
//                                V(i+1) = B^T d B from the raw block of chunk i+1         <- VALU + LDS, in the MFMA shadows
//                                U(i+1) = G g G^T from the filter taps loaded at the top of the iteration
//                                raw block of chunk i+2: global loads at the top, LDS stores at the bottom
//
// U, V and the raw block are double-buffered (2 x 32 KB + 2 x 32 KB + 2 x 13.75 KB = 155.5 KB of the CU's 160 KB LDS, one
// workgroup per CU), and the workgroups are persistent: the chunk sequence runs on across tile blocks, so a block's
// prologue hides under the previous block's MFMAs and only the output transform + stores stay exposed.
//
// LDS layouts (no padding -- a rotation keeps every access conflict-free):
//   U / V plane of one Winograd point xi: 8 channels x 64 (couts | tiles) floats; channel k = 4 ks + kq of (cout | tile) m
//   lives at  kq * 128 + ((2 m + ks + 32 kq) & 127):  the two k-steps of an MFMA lane are one 8-byte read, the 32 lanes of a
//   half-wave (16 m x 2 kq) cover 64 distinct banks, and the transform writes (lane = (ks, 32 m)) are 64 consecutive words.
//   raw block: channel planes [8][18 rows][24] + 8 floats between planes (planes 4 apart sit 32 banks apart).
#include "mrdis_common.h"
#include "mrdis_wino4.h"

struct Wino2Params {
    const float* in; const float* w; const float* bias; float* out;
    int N, H, W, Cin, ldin, Cout, ldout;
    int flip, lrelu, nt_out;
    int nby, nbx, coTiles, nblk;      // 8x8-tile blocks per image, 64-cout tiles, blocks in total
    unsigned in_bytes, w_bytes;       // record counts of the buffer descriptors
    // SPADE epilogue (wino2_kernel<.., true>): the filter is the fused gamma | beta filter of a SPADE block (Cout = 2 C); a workgroup owns
    // 32 channels and their 32 gamma + 32 beta couts, a lane gets gamma and beta of the same 4 channels, and the epilogue writes
    // out = (z - mean) * rstd * (1 + gamma) + beta and gamma itself (model.py:2440-2446) -- the 2C-channel tensor never exists
    const float* z; const float* mean; const float* rstd; float* gamma_out;
    int ldz, ldg, C;
    // UIMG: the filter already in the Winograd domain (mrdis_wino_u_jobs): [cout tile][chunk][8 channels][4 point groups][64 couts][4 points],
    // zero where the chunk / the tile runs past Cin / Cout -- what U = G g G^T of this kernel's (chunk, cout tile) walk needs, in its order
    const float* u_img; unsigned u_bytes;
};

namespace {
constexpr int KC = 8, NT = 512, RHW = 18, RWP = 24, PIXP = RHW * RWP + 8;
constexpr int XI = 512;                               // floats of one Winograd point's U or V plane
constexpr int UVBUF = 16 * XI, RAWBUF = KC * PIXP;
constexpr int BIASBUF = 1024;                          // the bias vector, staged once (a global load in the epilogue costs its full latency)
constexpr size_t WINO2_LDS = sizeof(float) * (4 * UVBUF + 2 * RAWBUF + BIASBUF);
template <int V_> struct IC { static constexpr int value = V_; };
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// An LDS pointer the compiler may not fold constants into: accesses at ptr[k * XI] then use the instruction's 16-bit offset
// field; left alone, hipcc builds one address register per 2 KB plane beyond the first 64 KB (27 spilled registers in wgrad2).
// (The laundering is done on the element INDEX: a laundered pointer would lose its LDS address space and turn into flat loads.)
__device__ __forceinline__ int w2_opaque(int idx) { asm volatile("" : "+v"(idx)); return idx; }
constexpr unsigned W2_OOB = 0xfffffff0u;               // byte offset past every record count: the buffer load returns zeros
}  // namespace

// ABL: timing-only ablations (results wrong): 1 no V transform, 2 no U transform, 4 no MFMAs, 8 no filter loads, 32 no raw loads, 16 no operand reads
// UIMG: U comes pre-transformed from the filter image (four 16-byte loads per thread and chunk, no G g G^T in the loop) instead of nine
// tap loads + the transform: the same 16 values (the image is built with the same expressions), ~40 VALU and 5 vector-memory
// instructions fewer per thread and chunk in the MFMA shadows.
template <int ABL, bool SPADE = false, bool UIMG = false>
__global__ __launch_bounds__(512, 1) void wino2_kernel(const Wino2Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ub = smem;                           // [2][16][XI]
    float* const Vb = smem + 2 * UVBUF;               // [2][16][XI]
    float* const Rb = smem + 4 * UVBUF;               // [2][KC][PIXP]
    float* const Bs = Rb + 2 * RAWBUF;                // [BIASBUF]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int cg = wave & 1, tg = wave >> 1;
    // transform role: one (channel, cout) filter pair and one (channel, tile) patch per chunk
    const int ks_t = lane & 1, idx_t = 32 * (wave >> 2) + (lane >> 1), kq_t = wave & 3;
    const int k_t = 4 * ks_t + kq_t;
    const int t_dst = kq_t * 128 + ((2 * idx_t + ks_t + 32 * kq_t) & 127);
    const int v_src = k_t * PIXP + (2 * (idx_t >> 3)) * RWP + 2 * (idx_t & 7);
    // MFMA role
    const int a0_off = kq * 128 + ((2 * (32 * cg + l16) + 32 * kq) & 127);
    const int a1_off = kq * 128 + ((2 * (32 * cg + 16 + l16) + 32 * kq) & 127);
    const int b_off = kq * 128 + ((2 * (16 * tg + l16) + 32 * kq) & 127);
    // staging role: two (pixel, 4-channel group) items of the 18 x 18 raw block
    int s_l[2], s_ry[2], s_rx[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * NT, pi = idx >> 1;
        s_ry[it] = pi / RHW; s_rx[it] = pi - s_ry[it] * RHW;
        s_l[it] = (idx < RHW * RHW * 2) ? 4 * (idx & 1) * PIXP + s_ry[it] * RWP + s_rx[it] : -1;
    }
    const int q4 = 4 * (tid & 1);

    const int grid = gridDim.x;
    const int rb = mrdis_xcd_remap(blockIdx.x, grid);
    const int nmine = (p.nblk - rb + grid - 1) / grid;            // host: grid <= nblk
    const int nch = (p.Cin + KC - 1) / KC;
    const int total = nmine * nch;
    auto decode = [&](int j, int& n, int& oy0, int& ox0, int& co0) {
        int b = rb + j * grid;
        const int cot = b % p.coTiles; b /= p.coTiles;
        const int bx = b % p.nbx; b /= p.nbx;
        const int by = b % p.nby;
        n = b / p.nby; oy0 = 16 * by; ox0 = 16 * bx; co0 = 64 * cot;
    };

    // ---- raw-block cursor (two iterations ahead of the MFMAs)
    // All global loads are buffer loads whose offset is W2_OOB where there is nothing to read (zeros come back): no branch
    // around a load, so the compiler's vmcnt bookkeeping stays exact and a wait covers only the loads it must.
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = UIMG ? __builtin_amdgcn_make_buffer_rsrc((void*)p.u_img, 0, p.u_bytes, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned xg[2] = {W2_OOB, W2_OOB};
    int rj = 0, rc = 0;
    auto raw_block = [&]() {
        xg[0] = W2_OOB; xg[1] = W2_OOB;
        if (rj < nmine) {
            int n, oy0, ox0, co0; decode(rj, n, oy0, ox0, co0);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int h = oy0 - 1 + s_ry[it], w_ = ox0 - 1 + s_rx[it];
                if (s_l[it] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                    xg[it] = 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldin + q4);      // host: < 2^30 elements
            }
        }
    };
    float4 xr[2][2];                                  // two register sets: a load has two iterations to land (HBM latency under load
                                                      // is longer than one iteration's 64 MFMAs)
    unsigned xo[2] = {W2_OOB, W2_OOB};                // byte offsets of the NEXT raw-block load
    auto raw_next = [&]() {                           // cursor bookkeeping apart from the loads (see filt_next)
        const int c0 = rc * KC;
        const bool c_ok = c0 + q4 < p.Cin;
#pragma unroll
        for (int it = 0; it < 2; ++it) xo[it] = (c_ok && xg[it] != W2_OOB) ? xg[it] + 4u * c0 : W2_OOB;
        if (++rc == nch) { rc = 0; ++rj; raw_block(); }
    };
    auto load_raw1 = [&](auto S_, int it) {
        constexpr int S = decltype(S_)::value;
        if (ABL & 32) { xr[S][it] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)xo[it], 0, 0);
        xr[S][it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto load_raw = [&](auto S_) { raw_next(); load_raw1(S_, 0); load_raw1(S_, 1); };
    auto raw_store = [&](auto S_, float* Rw) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (s_l[it] >= 0) { float* d_ = Rw + s_l[it]; d_[0] = xr[S][it].x; d_[PIXP] = xr[S][it].y; d_[2 * PIXP] = xr[S][it].z; d_[3 * PIXP] = xr[S][it].w; }
    };

    // ---- filter cursor (one iteration ahead)
    int fj = 0, fc = 0, f_co = 0; bool f_on = false;
    unsigned f_tile = 0;                              // UIMG: byte offset of this thread's first 16 bytes inside chunk 0 of the block's cout tile
    auto filt_block = [&]() {
        f_on = false;
        if (fj < nmine) {
            int n, oy0, ox0, co0; decode(fj, n, oy0, ox0, co0);
            if (UIMG) { f_on = true; f_tile = 4u * (unsigned)(((co0 >> 6) * nch) * (16 * XI) + (k_t * 4 * 64 + idx_t) * 4); return; }
            if (SPADE) {      // 16-cout blocks of the workgroup: gamma[c0..+15], beta[c0..+15], gamma[c0+16..+31], beta[c0+16..+31]
                const int ch = co0 / 2 + 16 * (idx_t >> 5) + (idx_t & 15);
                f_co = ((idx_t >> 4) & 1 ? p.C : 0) + ch; f_on = ch < p.C;
            } else { f_co = co0 + idx_t; f_on = f_co < p.Cout; }
        }
    };
    const unsigned tstride4 = 4u * (unsigned)(p.Cin * p.Cout);
    float gr[9];
    unsigned f_wo = W2_OOB;                           // byte offset of tap 0 of this thread's (channel, cout) pair for the NEXT filter load
    auto filt_next = [&]() {                          // cursor bookkeeping (branches, divisions) apart from the loads themselves
        if (UIMG) f_wo = f_on ? f_tile + 4u * (unsigned)(fc * 16 * XI) : W2_OOB;
        else {
        const int k = ((ABL & 64) ? 0 : fc * KC) + k_t;      // 64: the same filter chunk every iteration (L1 hits)
        f_wo = (f_on && k < p.Cin) ? 4u * (unsigned)(k * p.Cout + f_co) : W2_OOB;
        }
        if (++fc == nch) { fc = 0; ++fj; filt_block(); }
    };
    float4 ur[4];                                     // UIMG: points 4 a .. 4 a + 3 of this thread's (channel, cout) pair, a = 0..3
    auto load_u = [&](int a) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(f_wo != W2_OOB ? f_wo + 1024u * (unsigned)a : W2_OOB), 0, 0);
        ur[a] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto u_put = [&](float* Un, int a) {
        float* up = Un + 4 * a * XI;
        up[0] = ur[a].x; up[XI] = ur[a].y; up[2 * XI] = ur[a].z; up[3 * XI] = ur[a].w;
    };
    auto load_tap = [&](int t) {
        gr[t] = (ABL & 8) ? 0.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_w, (int)(f_wo != W2_OOB ? f_wo + (unsigned)(p.flip ? 8 - t : t) * tstride4 : W2_OOB), 0, 0));
    };
    auto load_filt = [&]() {
#pragma unroll
        for (int t = 0; t < 9; ++t) load_tap(t);
    };

    // ---- transform pieces (called whole in the prologue, one piece per MFMA step in the loop)
    float d[4][4], r[4][4], t_[4][3];
    auto v_row = [&](const float* Rr, int i) {
        const float2 lo = *reinterpret_cast<const float2*>(Rr + i * RWP), hi = *reinterpret_cast<const float2*>(Rr + i * RWP + 2);
        d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
    };
    auto v_col = [&](int j) {                         // B^T d: B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        r[0][j] = d[0][j] - d[2][j]; r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j]; r[3][j] = d[1][j] - d[3][j];
    };
    auto v_out = [&](float* Vn, int a) {
        float* vp = Vn + 4 * a * XI;                  // Vn / Un include this thread's t_dst, Rr its v_src
        vp[0] = r[a][0] - r[a][2]; vp[XI] = r[a][1] + r[a][2]; vp[2 * XI] = r[a][2] - r[a][1]; vp[3 * XI] = r[a][1] - r[a][3];
    };
    auto u_col = [&]() {                              // G g: G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float g0 = gr[j], g1 = gr[3 + j], g2 = gr[6 + j];
            t_[0][j] = g0; t_[1][j] = 0.5f * (g0 + g1 + g2); t_[2][j] = 0.5f * (g0 - g1 + g2); t_[3][j] = g2;
        }
    };
    auto u_out = [&](float* Un, int a) {
        float* up = Un + 4 * a * XI;
        up[0] = t_[a][0]; up[XI] = 0.5f * (t_[a][0] + t_[a][1] + t_[a][2]);
        up[2 * XI] = 0.5f * (t_[a][0] - t_[a][1] + t_[a][2]); up[3 * XI] = t_[a][2];
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- prologue: U(0), V(0), raw(1) in LDS, raw(2) and filter(1) in register set 1; cursors at raw(3), filter(2)
    for (int c = threadIdx.x; c < BIASBUF; c += NT) Bs[c] = (p.bias != nullptr && c < p.Cout) ? p.bias[c] : 0.f;
    raw_block(); filt_block();
    load_raw(IC<0>{}); filt_next();
    if (UIMG) {
#pragma unroll
        for (int a = 0; a < 4; ++a) load_u(a);
    } else load_filt();
    load_raw(IC<1>{});
    raw_store(IC<0>{}, Rb);
    if (UIMG) {
#pragma unroll
        for (int a = 0; a < 4; ++a) u_put(Ub + t_dst, a);
        filt_next();
#pragma unroll
        for (int a = 0; a < 4; ++a) load_u(a);        // U(1), as if issued in steps 11-14 of an iteration -1
    } else {
    u_col();
#pragma unroll
    for (int a = 0; a < 4; ++a) u_out(Ub + t_dst, a);
    filt_next(); load_filt();                         // filter(1), as if issued in step 11 of an iteration -1
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) v_row(Rb + v_src, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) v_col(j);
#pragma unroll
    for (int a = 0; a < 4; ++a) v_out(Vb + t_dst, a);
    raw_store(IC<1>{}, Rb + RAWBUF);
    load_raw(IC<1>{});
    __syncthreads();

    int mj = 0, mc = 0;
    // one iteration; P = its parity = the U / V / raw buffers and the register set it LOADS into
    auto iteration = [&](auto P_) {
        constexpr int P = decltype(P_)::value;
        const float* Ua0 = smem + w2_opaque(P * UVBUF + a0_off); const float* Ua1 = smem + w2_opaque(P * UVBUF + a1_off);
        const float* Vbv = smem + w2_opaque((2 + P) * UVBUF + b_off);
        float* Un = smem + w2_opaque((P ^ 1) * UVBUF + t_dst); float* Vn = smem + w2_opaque((2 + (P ^ 1)) * UVBUF + t_dst);
        const float* Rr = smem + w2_opaque(4 * UVBUF + (P ^ 1) * RAWBUF + v_src);      // raw block of chunk i + 1
        float* Rw = Rb + P * RAWBUF;                  // raw block of chunk i + 2 goes where chunk i's was
        // Global loads go out ONE OR TWO PER STEP: a burst of 11 vector-memory instructions per wave fills the CU's memory
        // queue, the in-order waves stall on their loads and the matrix pipe idles behind them (measured: 65 us of a 375 us
        // launch for the nine filter loads issued back to back, 30 us for the two raw loads).
        filt_next();                                  // offsets of filter(i + 2): loaded in steps 11-15, once U(i + 1) has left `gr`
        raw_next();                                   // offsets of raw(i + 3): loaded in steps 2 and 6

        float2 a0[3], a1[3], bv[3];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            a0[s_] = *reinterpret_cast<const float2*>(Ua0 + s_ * XI);
            a1[s_] = *reinterpret_cast<const float2*>(Ua1 + s_ * XI);
            bv[s_] = *reinterpret_cast<const float2*>(Vbv + s_ * XI);
        }
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            if (s_ + 2 < 16 && !(ABL & 16)) {
                a0[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Ua0 + (s_ + 2) * XI);
                a1[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Ua1 + (s_ + 2) * XI);
                bv[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Vbv + (s_ + 2) * XI);
            }
            // the slice of the next chunk's transforms that rides in this step's MFMA shadows
            if (s_ == 2) load_raw1(IC<P>{}, 0);
            if (s_ == 6) load_raw1(IC<P>{}, 1);
            if (s_ < 4) { if (!(ABL & 1)) v_row(Rr, s_); }
            else if (s_ == 4) { if (!(ABL & 1)) { v_col(0); v_col(1); } }
            else if (s_ == 5) { if (!(ABL & 1)) { v_col(2); v_col(3); } }
            else if (s_ < 10) { if (!(ABL & 1)) v_out(Vn, s_ - 6); }
            else if (s_ == 10) { if (!UIMG && !(ABL & 2)) u_col(); }
            else if (s_ < 15) {
                if (UIMG) { u_put(Un, s_ - 11); load_u(s_ - 11); }       // U(i + 1) to LDS, then the same registers take U(i + 2)
                else { load_tap(2 * (s_ - 11)); load_tap(2 * (s_ - 11) + 1); if (!(ABL & 2)) u_out(Un, s_ - 11); }
            }
            else { if (!UIMG) load_tap(8); raw_store(IC<P ^ 1>{}, Rw); }
            const int c_ = s_ % 3;
            if (!(ABL & 4)) {
                acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c_].x, bv[c_].x, acc[s_][0], 0, 0, 0);
                acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c_].x, bv[c_].x, acc[s_][1], 0, 0, 0);
                acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c_].y, bv[c_].y, acc[s_][0], 0, 0, 0);
                acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c_].y, bv[c_].y, acc[s_][1], 0, 0, 0);
            } else { acc[s_][0][0] += a0[c_].x * bv[c_].x + a0[c_].y * bv[c_].y; acc[s_][1][0] += a1[c_].x * bv[c_].y; }
            // interleave the step's fillers with its MFMAs (left alone hipcc issues one MFMA, all fillers, then three MFMAs back to back)
#pragma unroll
            for (int g_ = 0; g_ < 4; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // then at most one vector-memory read,
                __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);      // two LDS operations
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // and three VALU instructions in its shadow
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();

        if (++mc == nch) {
            // ---- epilogue of block mj: lane = tile (16 tg + l16), couts co0 + 32 cg + 16 b + 4 kq + r; Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]
            int n, oy0, ox0, co0; decode(mj, n, oy0, ox0, co0);
            mc = 0; ++mj;
            const int tile = 16 * tg + l16;
            const int oy = oy0 + 2 * (tile >> 3), ox = ox0 + 2 * (tile & 7);
            if (SPADE) {
                // lane: channels ch .. ch + 3; accumulator block 0 = their gamma, block 1 = their beta
                const int ch = co0 / 2 + 16 * cg + 4 * kq;
                const bool ch_ok = ch < p.C;
                // the loads first: their latency hides under the output transform
                f32x4 zv[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const bool ok = ch_ok && oy + dy < p.H && ox + dx < p.W;
                        zv[dy][dx] = ok ? *reinterpret_cast<const f32x4*>(p.z + ((long long)(n * p.H + oy + dy) * p.W + ox + dx) * p.ldz + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                const f32x4 mu = ch_ok ? *reinterpret_cast<const f32x4*>(p.mean + (long long)n * p.C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 rs = ch_ok ? *reinterpret_cast<const f32x4*>(p.rstd + (long long)n * p.C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 y[2][2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x4 t0[4], t1[4];
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2) {
                        t0[i2] = acc[4 * i2][b] + acc[4 * i2 + 1][b] + acc[4 * i2 + 2][b];
                        t1[i2] = acc[4 * i2 + 1][b] - acc[4 * i2 + 2][b] - acc[4 * i2 + 3][b];
                    }
                    y[b][0][0] = t0[0] + t0[1] + t0[2]; y[b][0][1] = t1[0] + t1[1] + t1[2];
                    y[b][1][0] = t0[1] - t0[2] - t0[3]; y[b][1][1] = t1[1] - t1[2] - t1[3];
#pragma unroll
                    for (int x = 0; x < 16; ++x) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const f32x4 bg = *reinterpret_cast<const f32x4*>(Bs + (ch_ok ? ch : 0));
                const f32x4 bb = *reinterpret_cast<const f32x4*>(Bs + (ch_ok ? p.C + ch : 0));
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        if (!(ch_ok && oy + dy < p.H && ox + dx < p.W)) continue;
                        const f32x4 g = y[0][dy][dx] + bg, bt = y[1][dy][dx] + bb;
                        const f32x4 o = (zv[dy][dx] - mu) * rs * (g + 1.f) + bt;
                        const long long pix = (long long)(n * p.H + oy + dy) * p.W + ox + dx;
                        *reinterpret_cast<f32x4*>(p.out + pix * p.ldout + ch) = o;
                        *reinterpret_cast<f32x4*>(p.gamma_out + pix * p.ldg + ch) = g;
                    }
                return;
            }
            const bool full = oy0 + 16 <= p.H && ox0 + 16 <= p.W && co0 + 64 <= p.Cout;        // block-uniform: no per-store tests
            const f32x4 slope = p.lrelu ? f32x4{0.2f, 0.2f, 0.2f, 0.2f} : f32x4{1.f, 1.f, 1.f, 1.f};
            float* const o00 = p.out + ((long long)(n * p.H + oy) * p.W + ox) * p.ldout + co0 + 32 * cg + 4 * kq;
            const long long rowp = (long long)p.W * p.ldout;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                // the four couts of a lane are the components of its accumulator vectors: the output transform is element-wise
                // on f32x4 (packed adds)
                f32x4 t0[4], t1[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    t0[i2] = acc[4 * i2][b] + acc[4 * i2 + 1][b] + acc[4 * i2 + 2][b];
                    t1[i2] = acc[4 * i2 + 1][b] - acc[4 * i2 + 2][b] - acc[4 * i2 + 3][b];
                }
                f32x4 y[2][2];
                y[0][0] = t0[0] + t0[1] + t0[2]; y[0][1] = t1[0] + t1[1] + t1[2];
                y[1][0] = t0[1] - t0[2] - t0[3]; y[1][1] = t1[1] - t1[2] - t1[3];
#pragma unroll
                for (int x = 0; x < 16; ++x) acc[x][b] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int co = co0 + 32 * cg + 16 * b + 4 * kq;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bs + co);
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        f32x4 v = y[dy][dx] + b4;
                        v = __builtin_elementwise_max(v, v * slope);              // LeakyReLU(0.2), or the identity
                        float* dst = o00 + dy * rowp + dx * p.ldout + 16 * b;
                        if ((ABL & 256) ? (v[0] == 1.2345f) : (full || (co < p.Cout && oy + dy < p.H && ox + dx < p.W))) {
                            if (p.nt_out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                            else *reinterpret_cast<f32x4*>(dst) = v;
                        }
                    }
            }
        }
    };
    for (int i = 0; i < total; i += 2) {
        iteration(IC<0>{});
        if (i + 1 < total) iteration(IC<1>{});
    }
}

// returns MRDIS_EUNSUPPORTED when the layer is outside what this kernel covers (the caller then takes wino_conv_kernel)
static unsigned wino_u_bytes(int R, int tiles) { return (unsigned)(4LL * tiles * mrdis_cdiv(R, KC) * 16 * XI); }

int mrdis_run_wino2(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int N, int H, int W, int Ci, int Co, int flip, int lrelu, hipStream_t s, const float* u_img) {
    if (Co <= 32 || Co > BIASBUF || Co % 4 != 0 || Ci % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x3fffffffLL || 9LL * Ci * Co >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    Wino2Params p{};
    p.in_bytes = (unsigned)(4LL * ((long long)(N * H) * W - 1) * ldx + 4LL * Ci); p.w_bytes = (unsigned)(36LL * Ci * Co);
    p.in = x; p.w = w; p.bias = bias; p.out = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldy;
    p.flip = flip; p.lrelu = lrelu;
    { const long long mb = mrdis_opt(MRDIS_OPT_NT_MB); p.nt_out = (long long)N * H * W * ldy * 4 >= mb * 1000000LL ? 1 : 0; }
    p.nby = mrdis_cdiv((H + 1) / 2, 8); p.nbx = mrdis_cdiv((W + 1) / 2, 8);
    p.coTiles = mrdis_cdiv(Co, 64);
    const long long nblk = (long long)N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.nblk = (int)nblk;
    if (!mrdis_opt(MRDIS_OPT_WINO_U) || (((uintptr_t)u_img) & 15) != 0) u_img = nullptr;
    p.u_img = u_img; p.u_bytes = wino_u_bytes(Ci, p.coTiles);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)wino2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO2_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        if (hipFuncSetAttribute((const void*)wino2_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO2_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
#ifdef WINO2_ABLATIONS
#define W2A(a) hipFuncSetAttribute((const void*)wino2_kernel<a>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO2_LDS);
        W2A(1) W2A(2) W2A(3) W2A(4) W2A(8) W2A(32) W2A(40) W2A(43) W2A(16) W2A(47) W2A(64) W2A(128) W2A(256)
#undef W2A
#endif
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = nblk < n_cu ? (int)nblk : n_cu;
#ifdef WINO2_ABLATIONS
    const int abl = (int)mrdis_opt(MRDIS_OPT_MODE);          // debug_mode doubles as the ablation selector in this build
#define W2A(a) if (abl == a) { MRDIS_LAUNCH(wino2_kernel<a>, dim3(grid), dim3(NT), WINO2_LDS, s, p); MRDIS_CHECK_LAUNCH(); return MRDIS_OK; }
    W2A(1) W2A(2) W2A(3) W2A(4) W2A(8) W2A(32) W2A(40) W2A(43) W2A(16) W2A(47) W2A(64) W2A(128) W2A(256)
#undef W2A
#endif
    mrdis_count(MRDIS_CNT_WINO2);
    if (u_img) MRDIS_LAUNCH((wino2_kernel<0, false, true>), dim3(grid), dim3(NT), WINO2_LDS, s, p);
    else MRDIS_LAUNCH(wino2_kernel<0>, dim3(grid), dim3(NT), WINO2_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// SPADE-fused form: x = si_out (N, H, W, Ci), w = fused gamma | beta filter [9][Ci][2 C], bias (2 C); z (N, H, W, C) with its instance
// statistics; writes mix = (z - mean) rstd (1 + gamma) + beta and gamma.  MRDIS_EUNSUPPORTED: the caller runs the two-step path.
int mrdis_run_wino2_spade(const float* x, int ldx, const float* w, const float* bias, const float* z, int ldz, const float* mean, const float* rstd,
                          float* mix, int ldmix, float* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s, const float* u_img) {
    const int Co = 2 * C;
    if (C < 16 || C % 16 != 0 || Co > BIASBUF || Ci % 4 != 0 || ldx % 4 != 0 || ldz % 4 != 0 || ldmix % 4 != 0 || ldg % 4 != 0) return MRDIS_EUNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)z) | ((uintptr_t)mix) | ((uintptr_t)gamma) | ((uintptr_t)mean) | ((uintptr_t)rstd)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x3fffffffLL || 9LL * Ci * Co >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    Wino2Params p{};
    p.in_bytes = (unsigned)(4LL * ((long long)(N * H) * W - 1) * ldx + 4LL * Ci); p.w_bytes = (unsigned)(36LL * Ci * Co);
    p.in = x; p.w = w; p.bias = bias; p.out = mix;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldmix;
    p.z = z; p.ldz = ldz; p.mean = mean; p.rstd = rstd; p.gamma_out = gamma; p.ldg = ldg; p.C = C;
    p.nby = mrdis_cdiv((H + 1) / 2, 8); p.nbx = mrdis_cdiv((W + 1) / 2, 8);
    p.coTiles = mrdis_cdiv(C, 32);
    const long long nblk = (long long)N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.nblk = (int)nblk;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)wino2_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO2_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        if (hipFuncSetAttribute((const void*)wino2_kernel<0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO2_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (!mrdis_opt(MRDIS_OPT_WINO_U) || (((uintptr_t)u_img) & 15) != 0) u_img = nullptr;
    p.u_img = u_img; p.u_bytes = wino_u_bytes(Ci, p.coTiles);
    const int grid = nblk < n_cu ? (int)nblk : n_cu;
    mrdis_count(MRDIS_CNT_WINO2_SPADE);
    if (u_img) MRDIS_LAUNCH((wino2_kernel<0, true, true>), dim3(grid), dim3(NT), WINO2_LDS, s, p);
    else MRDIS_LAUNCH((wino2_kernel<0, true>), dim3(grid), dim3(NT), WINO2_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// =========================================================================== filter images for wino2_kernel<.., UIMG>
// One job = one (filter, role): w is [9][R][S] (forward: w_tck with R = Ci, S = Co; data gradient: w_tkc with R = Co, S = Ci and flip = 1;
// fused gamma | beta filter of a SPADE block: R = Ci, S = 2 C, spadeC = C -- the cout order of wino2_kernel<.., SPADE>).  A thread builds the
// sixteen points of one (reduction channel, cout slot) pair with the expressions of the in-kernel transform (u_col / u_out), so the
// pipelined kernel computes the same values either way (option wino_u = 0 / 1 is bit-identical).
struct WinoUJob { const float* w; float* img; int R, S, flip, spadeC, block0, nblk, fmt, pad_; };   // fmt: 2 = the image above, 4 = mrdis_wino4.h

__global__ __launch_bounds__(256) void wino_u_jobs_kernel(const WinoUJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;                       // last job with block0 <= blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const WinoUJob j = jobs[lo];
    if (j.fmt == 5) {
        // narrow F(4x4, 3x3) image (mrdis_wino4.h): 32-cout tiles
        const int nch4 = (j.R + MRDIS_W4_KC - 1) / MRDIS_W4_KC, tiles4 = (j.S + 31) / 32;
        const long long total4 = (long long)tiles4 * nch4 * MRDIS_W4_KC * 32;
        for (long long i = ((long long)blockIdx.x - j.block0) * 256 + threadIdx.x; i < total4; i += (long long)j.nblk * 256) {
            const int m = (int)(i & 31), kq = (int)((i >> 5) & 3);
            const long long tc = i >> 7;
            const int c = (int)(tc % nch4), cot = (int)(tc / nch4);
            const int r = c * MRDIS_W4_KC + kq, co = cot * 32 + m;
            const bool ok = r < j.R && co < j.S;
            float gr[9], U[36];
#pragma unroll
            for (int t = 0; t < 9; ++t) gr[t] = ok ? j.w[((long long)(j.flip ? 8 - t : t) * j.R + r) * j.S + co] : 0.f;
            mrdis_w4_filter_transform(gr, U);
            float* dst = j.img + tc * MRDIS_W4N_UCHUNK + kq * 64;
#pragma unroll
            for (int pt = 0; pt < 36; ++pt) dst[(pt >> 1) * 256 + ((2 * m + (pt & 1) + 32 * kq) & 63)] = U[pt];
        }
        return;
    }
    if (j.fmt == 4) {
        // F(4x4, 3x3) image (mrdis_wino4.h): one thread = one (reduction channel, cout slot) pair = 36 values
        const int nch4 = (j.R + MRDIS_W4_KC - 1) / MRDIS_W4_KC, tiles4 = j.spadeC ? (j.spadeC + 31) / 32 : (j.S + 63) / 64;
        const long long total4 = (long long)tiles4 * nch4 * MRDIS_W4_KC * 64;
        for (long long i = ((long long)blockIdx.x - j.block0) * 256 + threadIdx.x; i < total4; i += (long long)j.nblk * 256) {
            const int m = (int)(i & 63), kq = (int)((i >> 6) & 3);
            const long long tc = i >> 8;              // cot * nch4 + chunk
            const int c = (int)(tc % nch4), cot = (int)(tc / nch4);
            const int r = c * MRDIS_W4_KC + kq;
            int co; bool ok;
            if (j.spadeC) {                           // slot m: 8-channel group m / 16, row m % 16: rows 0-7 gamma, rows 8-15 beta of the group's channels
                const int ch = cot * 32 + 8 * (m >> 4) + (m & 7);
                co = ((m & 8) ? j.spadeC : 0) + ch; ok = ch < j.spadeC;
            } else { co = cot * 64 + m; ok = co < j.S; }
            ok = ok && r < j.R;
            float gr[9], U[36];
#pragma unroll
            for (int t = 0; t < 9; ++t) gr[t] = ok ? j.w[((long long)(j.flip ? 8 - t : t) * j.R + r) * j.S + co] : 0.f;
            mrdis_w4_filter_transform(gr, U);
            float* dst = j.img + tc * MRDIS_W4_UCHUNK + kq * 128;
#pragma unroll
            for (int pt = 0; pt < 36; ++pt) dst[(pt >> 1) * MRDIS_W4_UPP + ((2 * m + (pt & 1) + 32 * kq) & 127)] = U[pt];
        }
        // ... followed by the 16-point image of the same filter: a call whose grid the F(4x4) kernel declines (small maps) runs the F(2x2) kernel on it
    }
    float* const img2 = j.fmt == 4 ? j.img + mrdis_wino4_image_floats(j.R, j.S, j.spadeC) : j.img;
    const int nch = (j.R + KC - 1) / KC;
    const int tiles = j.spadeC ? (j.spadeC + 31) / 32 : (j.S + 63) / 64;
    const long long total = (long long)tiles * nch * KC * 64;
    for (long long i = ((long long)blockIdx.x - j.block0) * 256 + threadIdx.x; i < total; i += (long long)j.nblk * 256) {
        const int slot = (int)(i & 63), k = (int)((i >> 6) & 7);
        const long long tc = i >> 9;                  // cot * nch + chunk
        const int c = (int)(tc % nch), cot = (int)(tc / nch);
        const int r = c * KC + k;
        int co; bool ok;
        if (j.spadeC) {
            const int ch = cot * 32 + 16 * (slot >> 5) + (slot & 15);
            co = (((slot >> 4) & 1) ? j.spadeC : 0) + ch; ok = ch < j.spadeC;
        } else { co = cot * 64 + slot; ok = co < j.S; }
        ok = ok && r < j.R;
        float gr[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) gr[t] = ok ? j.w[((long long)(j.flip ? 8 - t : t) * j.R + r) * j.S + co] : 0.f;
        float t_[4][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {                 // G g: G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
            const float g0 = gr[q], g1 = gr[3 + q], g2 = gr[6 + q];
            t_[0][q] = g0; t_[1][q] = 0.5f * (g0 + g1 + g2); t_[2][q] = 0.5f * (g0 - g1 + g2); t_[3][q] = g2;
        }
        float* dst = img2 + ((tc * KC + k) * 4) * 256 + slot * 4;       // [tc][k][a][slot][4]
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *reinterpret_cast<float4*>(dst + a * 256) = make_float4(t_[a][0], 0.5f * (t_[a][0] + t_[a][1] + t_[a][2]),
                                                                   0.5f * (t_[a][0] - t_[a][1] + t_[a][2]), t_[a][2]);
    }
}

extern "C" size_t mrdis_wino_u_job_bytes(void) { return sizeof(WinoUJob); }
static long long wino_u_elems(int R, int S, int spadeC) {       // (reduction channel, cout slot) pairs of the 16-point image
    const int tiles = spadeC ? (spadeC + 31) / 32 : (S + 63) / 64;
    return (long long)tiles * ((R + KC - 1) / KC) * KC * 64;
}
extern "C" int mrdis_wino_u_format(int R, int S, int spadeC) { return mrdis_wino_u_fmt(R, S, spadeC); }
extern "C" long long mrdis_wino_u_image_floats_fmt(int R, int S, int spadeC, int fmt) {
    if (!mrdis_wino_u_fmt_valid(R, S, spadeC, fmt)) return -1;
    if (fmt == 5) return mrdis_wino4n_image_floats(R, S);
    return (fmt == 4 ? mrdis_wino4_image_floats(R, S, spadeC) : 0) + 16 * wino_u_elems(R, S, spadeC);
}
extern "C" long long mrdis_wino_u_image_floats(int R, int S, int spadeC) { return mrdis_wino_u_image_floats_fmt(R, S, spadeC, mrdis_wino_u_fmt(R, S, spadeC)); }
extern "C" int mrdis_wino_u_job_blocks(int R, int S, int spadeC) {
    const long long b = (wino_u_elems(R, S, spadeC) + 255) / 256;
    const int cap = mrdis_wino_u_fmt(R, S, spadeC) >= 4 ? 128 : 64;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}
extern "C" int mrdis_wino_u_jobs(const void* jobs, int njobs, int total_blocks, void* stream) {
    if (!jobs || njobs < 1 || total_blocks < njobs) return MRDIS_EINVAL;
    MRDIS_LAUNCH(wino_u_jobs_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const WinoUJob*>(jobs), njobs);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// =========================================================================== pipelined Winograd weight gradient
// wino_wgrad_kernel<4, 2> (mrdis_wino.hip) with the same pipeline: a workgroup owns a 64 x 64 block of (ci, co) and every
// `splits`-th block of 2 x 4 tiles; iteration i multiplies V(i) [16 points][8 tiles][64 ci] by Z(i) [16][8][64 co] (64 MFMAs per
// wave, reduction axis = tile) while the same waves build V(i+1) = B^T d B from the raw x block of iteration i+1 (LDS) and
// Z(i+1) = A dY A^T from the four dy values loaded one iteration earlier, store the raw block of iteration i+2 and issue the
// loads of iterations i+3 (x) and i+2 (dy), one per MFMA step.  V / Z planes use the rotation layout of wino2_kernel (the tile
// index is the MFMA k axis); the raw block is [pixel][64 ci] with 8 floats of skew per block row, so the transform's reads
// (lanes = 16 ci x 2 tile rows) and the 16-byte staging writes are conflict-free.  2 x 32 + 2 x 32 + 2 x 15.2 KB = 158.4 KB of LDS.
#include "mrdis_wino.h"

namespace {
constexpr int G_RH = 6, G_RW = 10, G_NPX = G_RH * G_RW;
constexpr int G_RAW = G_NPX * 64 + G_RH * 8;            // floats of one raw x block
constexpr size_t WGRAD2_LDS = sizeof(float) * (4 * UVBUF + 2 * G_RAW);
__device__ __forceinline__ int g_raw_off(int ry, int rx) { return (ry * G_RW + rx) * 64 + ry * 8; }
}  // namespace

__global__ __launch_bounds__(512, 1) void wino_wgrad2_kernel(const WinoWgradParams p, const int s_n, const int s_by, const int s_bx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Vb = smem;                           // [2][16][XI]
    float* const Zb = smem + 2 * UVBUF;               // [2][16][XI]
    float* const Rb = smem + 4 * UVBUF;               // [2][G_RAW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
    const int wi = wave & 3, wo = wave >> 2;
    int b_ = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);   // the (cib, cob) workgroups of a split read the same x / dy tiles: same XCD, same L2
    const int cob = b_ % p.nCoB; b_ /= p.nCoB;
    const int cib = b_ % p.nCiB;
    const int split = b_ / p.nCiB;
    const int ci0 = cib * 64, co0 = cob * 64;

    // transform role: tile = 4 ks + kq_t (tile row ks, tile column kq_t of the 2 x 4 block), channel m of the 64 (ci for V, co for Z)
    const int ks_t = lane & 1, m_t = 32 * (wave >> 2) + (lane >> 1), kq_t = wave & 3;
    const int t_dst = kq_t * 128 + ((2 * m_t + ks_t + 32 * kq_t) & 127);
    const int v_src = g_raw_off(2 * ks_t, 2 * kq_t) + m_t;
    // MFMA role: A = V (rows ci = 16 wi + l16), B = Z (columns co = 32 wo + 16 b + l16), k = tile
    const int a_off = kq * 128 + ((2 * (16 * wi + l16) + 32 * kq) & 127);
    const int b0_off = kq * 128 + ((2 * (32 * wo + l16) + 32 * kq) & 127);
    const int b1_off = kq * 128 + ((2 * (32 * wo + 16 + l16) + 32 * kq) & 127);
    // staging role: two (pixel, 4-channel group) items of the 6 x 10 raw block (960 items)
    int s_l[2], s_rc[2];                              // LDS offset (or -1), (row << 8) | column in the raw block
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * NT, pi = idx >> 4, ry = pi / G_RW, rx = pi - ry * G_RW;
        s_rc[it] = (ry << 8) | rx;
        s_l[it] = (idx < G_NPX * 16) ? g_raw_off(ry, rx) + 4 * (idx & 15) : -1;
    }
    const int q4 = 4 * (tid & 15);

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)(4LL * ((long long)(p.N * p.H) * p.W - 1) * p.ldx + 4LL * p.Ci), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)(4LL * ((long long)(p.N * p.H) * p.W - 1) * p.lddy + 4LL * p.Co), 0x00020000);

    // block cursors: (n, by, bx) advance by `splits` blocks per iteration = (s_n, s_by, s_bx) with carries (wave-uniform, scalar)
    struct Cur { int n, by, bx; };
    auto advance = [&](Cur& c) {
        c.bx += s_bx; if (c.bx >= p.nbx) { c.bx -= p.nbx; ++c.by; }
        c.by += s_by; if (c.by >= p.nby) { c.by -= p.nby; ++c.n; }
        c.n += s_n;
    };
    Cur rcur, dcur;
    { int t = split; rcur.bx = t % p.nbx; t /= p.nbx; rcur.by = t % p.nby; rcur.n = t / p.nby; dcur = rcur; }

    // Offsets are computed in the step that issues the load (short live ranges: the kernel sits at the 256-register limit)
    float4 xr[2][2];
    auto load_raw1 = [&](auto S_, int it) {           // item `it` of the raw block of the cursor's block; the cursor advances after item 1
        constexpr int S = decltype(S_)::value;
        const int h = 4 * rcur.by - 1 + (s_rc[it] >> 8), w_ = 8 * rcur.bx - 1 + (s_rc[it] & 255);
        const bool ok = s_l[it] >= 0 && rcur.n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
        const unsigned o = ok ? 4u * (unsigned)(((rcur.n * p.H + h) * p.W + w_) * p.ldx + ci0 + q4) : W2_OOB;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)o, 0, 0);
        xr[S][it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        if (it == 1) advance(rcur);
    };
    auto load_raw = [&](auto S_) { load_raw1(S_, 0); load_raw1(S_, 1); };
    auto raw_store = [&](auto S_, float* Rw) {
        constexpr int S = decltype(S_)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (s_l[it] >= 0) *reinterpret_cast<float4*>(Rw + s_l[it]) = xr[S][it];
    };

    float dr[4];
    auto load_dy1 = [&](int e) {                      // dy(row a, column b) of this thread's tile in the cursor's block; advances after the fourth
        const int a = e >> 1, b = e & 1;
        const int oy = 4 * dcur.by + 2 * ks_t + a, ox = 8 * dcur.bx + 2 * kq_t + b;
        const bool ok = dcur.n < p.N && oy < p.H && ox < p.W;
        const unsigned o = ok ? 4u * (unsigned)(((dcur.n * p.H + oy) * p.W + ox) * p.lddy + co0 + m_t) : W2_OOB;
        dr[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dy, (int)o, 0, 0));
        if (e == 3) advance(dcur);
    };

    float d[4][4], r[4][4], tz[4][2], bsum = 0.f;
    auto v_row = [&](const float* Rr, int i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = Rr[g_raw_off(i, j)];      // Rr includes this thread's v_src
    };
    auto v_col = [&](int j) {
        r[0][j] = d[0][j] - d[2][j]; r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j]; r[3][j] = d[1][j] - d[3][j];
    };
    auto v_out = [&](float* Vn, int a) {             // Vn / Zn include this thread's t_dst
        float* vp = Vn + 4 * a * XI;
        vp[0] = r[a][0] - r[a][2]; vp[XI] = r[a][1] + r[a][2]; vp[2 * XI] = r[a][2] - r[a][1]; vp[3 * XI] = r[a][1] - r[a][3];
    };
    auto z_col = [&]() {                              // A dY: A = [1 0; 1 1; 1 -1; 0 -1]; the bias gradient (sum of dy) rides along
        const float d00 = dr[0], d01 = dr[1], d10 = dr[2], d11 = dr[3];
        bsum += (d00 + d01) + (d10 + d11);
        tz[0][0] = d00; tz[0][1] = d01; tz[1][0] = d00 + d10; tz[1][1] = d01 + d11;
        tz[2][0] = d00 - d10; tz[2][1] = d01 - d11; tz[3][0] = -d10; tz[3][1] = -d11;
    };
    auto z_out = [&](float* Zn, int i) {
        float* zp = Zn + 4 * i * XI;
        zp[0] = tz[i][0]; zp[XI] = tz[i][0] + tz[i][1]; zp[2 * XI] = tz[i][0] - tz[i][1]; zp[3 * XI] = -tz[i][1];
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x) { acc[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- prologue: V(0), Z(0), raw(1) in LDS, raw(2) in register set 1, dy(1) in `dr`; cursors at raw(3), dy(2)
    load_raw(IC<0>{});
#pragma unroll
    for (int e = 0; e < 4; ++e) load_dy1(e);
    load_raw(IC<1>{});
    raw_store(IC<0>{}, Rb);
    z_col();
#pragma unroll
    for (int a = 0; a < 4; ++a) z_out(Zb + t_dst, a);
#pragma unroll
    for (int e = 0; e < 4; ++e) load_dy1(e);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) v_row(Rb + v_src, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) v_col(j);
#pragma unroll
    for (int a = 0; a < 4; ++a) v_out(Vb + t_dst, a);
    raw_store(IC<1>{}, Rb + G_RAW);
    load_raw(IC<1>{});
    __syncthreads();

    const int niter = (p.nblocks - split + p.splits - 1) / p.splits;
    auto iteration = [&](auto P_) {
        constexpr int P = decltype(P_)::value;
        const float* Va = smem + w2_opaque(P * UVBUF + a_off);
        const float* Zb0 = smem + w2_opaque((2 + P) * UVBUF + b0_off); const float* Zb1 = smem + w2_opaque((2 + P) * UVBUF + b1_off);
        float* Vn = smem + w2_opaque((P ^ 1) * UVBUF + t_dst); float* Zn = smem + w2_opaque((2 + (P ^ 1)) * UVBUF + t_dst);
        const float* Rr = smem + w2_opaque(4 * UVBUF + (P ^ 1) * G_RAW + v_src);     // raw block of iteration i + 1
        float* Rw = Rb + P * G_RAW;                   // raw block of iteration i + 2
        // raw(i + 3) is loaded in steps 1 and 5, dy(i + 2) in steps 11-14, once Z(i + 1) has left `dr`
        float2 av[3], b0[3], b1[3];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            av[s_] = *reinterpret_cast<const float2*>(Va + s_ * XI);
            b0[s_] = *reinterpret_cast<const float2*>(Zb0 + s_ * XI);
            b1[s_] = *reinterpret_cast<const float2*>(Zb1 + s_ * XI);
        }
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            if (s_ + 2 < 16) {
                av[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Va + (s_ + 2) * XI);
                b0[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Zb0 + (s_ + 2) * XI);
                b1[(s_ + 2) % 3] = *reinterpret_cast<const float2*>(Zb1 + (s_ + 2) * XI);
            }
            if (s_ == 1) load_raw1(IC<P>{}, 0);
            if (s_ == 5) load_raw1(IC<P>{}, 1);
            if (s_ < 4) v_row(Rr, s_);
            else if (s_ == 4) { v_col(0); v_col(1); }
            else if (s_ == 5) { v_col(2); v_col(3); }
            else if (s_ < 10) v_out(Vn, s_ - 6);
            else if (s_ == 10) z_col();
            else if (s_ < 15) { load_dy1(s_ - 11); z_out(Zn, s_ - 11); }
            else raw_store(IC<P ^ 1>{}, Rw);
            const int c_ = s_ % 3;
            acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].x, b0[c_].x, acc[s_][0], 0, 0, 0);
            acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].x, b1[c_].x, acc[s_][1], 0, 0, 0);
            acc[s_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].y, b0[c_].y, acc[s_][0], 0, 0, 0);
            acc[s_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].y, b1[c_].y, acc[s_][1], 0, 0, 0);
#pragma unroll
            for (int g_ = 0; g_ < 4; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    for (int i = 0; i < niter; i += 2) {
        iteration(IC<0>{});
        if (i + 1 < niter) iteration(IC<1>{});
    }

    // ---- epilogue: dg = G^T dU G per lane; D rows (4 kq + r) = ci, col l16 = co
    float* out = p.slab + (long long)split * 9 * p.Ci * p.Co;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int co = co0 + 32 * wo + 16 * b + l16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = ci0 + 16 * wi + 4 * kq + q;
            float t_[3][4];                           // rows: G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u0 = acc[j][b][q], u1 = acc[4 + j][b][q], u2 = acc[8 + j][b][q], u3 = acc[12 + j][b][q];
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
        // the eight threads (ks, kq_t) that hold channel m of this workgroup's 64 couts meet in LDS, fixed order
        float* red = smem;
        red[tid] = bsum;                              // every wave is past its last MFMA read of this region: the loop ends with a barrier
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) t += red[(4 * (tid >> 5) + k) * 64 + 2 * (tid & 31) + e];
            p.bias_slab[(long long)split * p.Co + co0 + tid] = t;
        }
    }
}

int mrdis_launch_wino_wgrad2(const WinoWgradParams& p, hipStream_t s) {
    if (p.D != 0 || p.Ci % 64 != 0 || p.Co % 64 != 0 || p.ldx % 4 != 0 || (((uintptr_t)p.x) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)p.N * p.H * p.W * p.ldx >= 0x3fffffffLL || (long long)p.N * p.H * p.W * p.lddy >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wino_wgrad2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WGRAD2_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        attr_set = true;
    }
    int t = p.splits;
    const int s_bx = t % p.nbx; t /= p.nbx;
    const int s_by = t % p.nby;
    const int s_n = t / p.nby;
    mrdis_count(MRDIS_CNT_WINO_WGRAD2);
    MRDIS_LAUNCH(wino_wgrad2_kernel, dim3(p.splits * p.nCiB * p.nCoB), dim3(NT), WGRAD2_LDS, s, p, s_n, s_by, s_bx);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
