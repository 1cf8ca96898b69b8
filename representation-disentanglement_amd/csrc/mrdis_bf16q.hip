// mrdis_bf16q.hip -- bf16 3x3 / stride 1 / pad 1 convolution on bf16 activations (MRDIS_DT_BF16), LDS-DMA form: forward, data gradient (flipped taps)
// and the SPADE-fused gamma | beta form of the layers with Cin a multiple of 32 on maps at least 32 wide.
//
// Why a second pipelined kernel.  Timing-only builds of bconv3_kernel (mrdis_bf16p.hip; tools/bconv_abl.py, 128 -> 256 at 64x64, B = 32: 94 us) said:
// without its MFMAs 97 us, without its LDS stores 57 us, without its global loads 66 us, the bare MFMA loop + epilogue 56 us.  The matrix pipe is not
// what it waits for; the LDS is: per 288 MFMAs a workgroup reads 432 KB of operands (a wave's 32 positions x 64 couts take 1.5 ds_read_b128 per MFMA) and stores
// 58 KB from registers (64 ds_write_b128 wave-instructions, whose VGPR -> LDS transfer is paced at 13 cycles each), ~95 % of the LDS cycles the MFMAs leave.
// Here:
//   * a wave owns 64 positions (two rows of a 16 x 32 tile) x all 64 couts of the workgroup: four MFMAs share two A and two B operand reads -- 1.0
//     ds_read_b128 per MFMA; the halo of the 512-position tile is 1.20x (8 x 32: 1.33x);
//   * the operand images reach LDS by DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write, no wait of the MFMA stream on a load; a lane
//     whose pixel / cout lies outside gets an offset beyond the descriptor (zeros: the convolution's padding, ragged tiles, cout tails);
//   * the images are UNPADDED 64-byte rows ([pixel][32 ch], [tap][cout][32 ch]) with the 16-byte piece p of row R stored at slot p ^ ((R >> 2) & 3): a DMA
//     wave-instruction fills 16 consecutive rows (lane l: row l / 4, slot l % 4, i.e. it fetches piece (l % 4) ^ ((R >> 2) & 3)), and the 16 lanes that one
//     ds_read_b128 cycle serves (lanes 0-3, 12-15, 20-27 of consecutive rows) touch all 64 banks once for any base row;
//   * two stages of (filter image 36 KB + pixel image 39 KB) are double-buffered: the DMA of item i + 1 is issued in the first ten steps of item i, every
//     wave waits for its own pieces (vmcnt(0)) before the ONE barrier per item.
// Arithmetic: the same products in the same accumulation order as bconv3_kernel / bconv_kernel (taps 0..8, two 16-channel k-steps per tap, 32-channel chunks in
// order): results are bit-identical to theirs.
#include "mrdis_tapconv.h"

typedef __bf16 bq_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bq_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned bq_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bq_u32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int Q_TW = 32, Q_INW = Q_TW + 2;
// geometry of one instantiation: NW waves per workgroup (each owns two rows of 32 positions), KC channels per item
template <int NW, int KC, int RPW = 2, int DMAW = NW> struct QGeo {
    static constexpr int TH = RPW * NW, INH = TH + 2, NPIX = INH * Q_INW;          // 16 x 32 tile: 18 x 34 = 612 halo pixels | 8 x 32 tile: 340
    static constexpr int RB = 2 * KC;                  // bytes of an image row (a pixel's / a cout's KC channels): 64 | 32
    static constexpr int PR = RB / 16;                 // 16-byte pieces per row: 4 | 2
    static constexpr int RPP = 1024 / RB;              // rows per DMA piece: 16 | 32
    static constexpr int XP = (NPIX + RPP - 1) / RPP;  // DMA pieces of a pixel image: 39 | 11
    static constexpr int XBYTES = XP * 1024;
    static constexpr int NXI = (XP + DMAW - 1) / DMAW; // per DMA wave: k = wave + DMAW i (k < XP)
    static constexpr int KS = KC / 16;                 // 16-channel k-steps per tap
    // swizzle of piece p of a row: slot = p ^ f(c), c = the row's index along the axis the lanes of a read walk (filter image: the cout row; pixel image: the
    // COLUMN of the pixel -- not its linear index, so that the tap rows of a lane differ by an immediate offset only).  f makes the 16 lanes that one ds_read_b128
    // cycle serves (lanes 0-3, 12-15, 20-27 | 4-11, 16-19, 28-31: consecutive c) hit all 64 banks once: 64-byte rows f = (c >> 2) & 3; 32-byte rows f = (c >> 3) & 1
    __device__ static __forceinline__ int f(int c) { return RB == 64 ? ((c >> 2) & 3) : ((c >> 3) & 1); }
};
constexpr int Q_BIAS = 1024;
constexpr unsigned Q_OOB = 0xfffffff0u;
template <int V_> struct QIC { static constexpr int value = V_; };
__device__ __forceinline__ int q_opaque(int idx) { asm volatile("" : "+v"(idx)); return idx; }
// see bp_pair8 (mrdis_bf16p.hip): the two half-waves of a position swap one 4-cout piece so that every lane holds 8 consecutive couts (one 16-byte store)
__device__ __forceinline__ bq_u32x4 bq_pair8(bq_u32x2 gq, bq_u32x2 gq1) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(gq[0], gq1[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(gq[1], gq1[1], false, false);
    return bq_u32x4{s0[0], s1[0], s0[1], s1[1]};
}
}  // namespace

struct BConv4Params {
    const void* in; const void* w; const float* bias; void* out;       // bf16 activations, bf16 filter [tap][Cout][Cin], fp32 bias
    int N, H, W, Cin, ldin, Cout, ldout;
    int tilesA, tilesB, coTiles, units;               // units = N * tilesA * tilesB * coTiles
    int nchunks, lrelu;
    int dh[9], dw[9], widx[9];
    unsigned in_bytes, w_bytes;
    // SPADE epilogue (see BConv3Params): Cout = 2 C, a workgroup owns 32 channels = 32 gamma + 32 beta couts
    const void* z; const float* mean; const float* rstd; void* gamma_out;
    int ldz, ldg, C;
    int wide;                                         // 16-byte output stores: Cout (C) % 8 == 0, output views 16-byte aligned with ld % 8 == 0
    int prio;                                         // 1: s_setprio 1 for waves 4-7 (the younger wave of every SIMD), 2: for waves 0-3, 0: none
    unsigned long long* dbg; int dbg_cap;             // diagnostic build (-DBCONV4_ABLATIONS, ABL & 64): per (workgroup < 4, wave) s_memtime stamps
};

// WC: 32-cout groups per workgroup (2: 64 couts, 1: the layers with <= 32 couts)
// ABL (timing-only, -DBCONV4_ABLATIONS): 1 no MFMAs, 2 no DMA after the prologue, 4 no output stores, 8 no wait for the DMA before the barrier
// NW, KC: 8 waves x 32-channel items (ONE workgroup per CU: 16 x 32 tiles) | 4 waves x 16-channel items (TWO independent workgroups per CU, 8 x 32 tiles: the
// epilogue, the DMA wait and the barrier of one run under the MFMAs of the other -- in-kernel stamps of the 8-wave form showed the two waves of a SIMD in lockstep,
// the older one done after ~4,000 cycles of an item and idle at the barrier for ~2,400 while the younger one needed ~6,100, and both in their epilogues together)
// RPW: rows of 32 positions per wave (2: two waves per SIMD share the matrix pipe | 4 with NW = 4: ONE wave per SIMD owns 128 positions x 64 couts, 8 MFMAs per 6 operand reads)
// FLIP: the tap table is the data gradient's (tap t reads pixel (1 - t / 3, 1 - t % 3) instead of (t / 3 - 1, t % 3 - 1)); p.widx[t] names the filter tap either way
// DMAW: the waves that issue the DMA (8: all | 4: waves 0-3 only -- the OLDER wave of every SIMD, which wins the matrix pipe's arbitration and would otherwise idle
// at the item's barrier while its partner catches up)
template <int WC, bool SPADE = false, int ABL = 0, int NW = 8, int KC = 32, int RPW = 2, bool FLIP = false, int DMAW = 4>
__global__ __launch_bounds__(64 * NW, 1) void bconv4_kernel(const BConv4Params p) {
    static_assert(!SPADE || WC == 2, "SPADE: 32 gamma + 32 beta couts per workgroup");
    using G = QGeo<NW, KC, RPW, DMAW>;
    constexpr int BN = 32 * WC, NT = 64 * NW;
    constexpr int Q_TH = G::TH, Q_NPIX = G::NPIX, Q_XP = G::XP, Q_XBYTES = G::XBYTES, Q_NXI = G::NXI, Q_KC = KC, RB = G::RB;
    constexpr int WROWS = 9 * BN;                      // rows of the filter image
    constexpr int WP = WROWS * RB / 1024;              // its DMA pieces (36 | 18 at 64-byte rows, 18 | 9 at 32-byte rows)
    constexpr int NWI = (WP + DMAW - 1) / DMAW;        // per DMA wave: k = wave + DMAW i (k < WP)
    constexpr int WBYTES = WROWS * RB;
    constexpr int XBASE = 2 * WBYTES;                  // LDS: [filter stage 0][filter stage 1][pixels stage 0][pixels stage 1][bias]
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_q[];
    float* const Bs = reinterpret_cast<float*>(smem_q + XBASE + 2 * Q_XBYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, e = lane & 31;
    // ---- MFMA role: positions (rows 2 wave + r, column e) of the 16 x 32 tile, k-group `half` (channels 8 half .. + 7 of a 16-channel k-step)
    // pixel (row RPW wave + r + 1 + dh, column e + 1 + dw) of the halo tile: a base per dw (the swizzle depends on the column only), rows as immediate offsets
    int xbase[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int col = e + d;                                            // e + 1 + dw, dw = d - 1
        xbase[d] = XBASE + ((RPW * wave) * Q_INW + col) * RB + ((half ^ G::f(col)) << 4);
    }
    auto tap_dh = [](int t) { return FLIP ? 1 - t / 3 : t / 3 - 1; };
    auto tap_dw = [](int t) { return FLIP ? 1 - t % 3 : t % 3 - 1; };
    const int aaddr = e * RB + ((half ^ G::f(e)) << 4);                  // filter row e of a 32-cout group: + (t * BN + 32 j) * RB, k-step 1: ^ 32 (the row offsets do not touch the swizzle bits)

    // ---- DMA roles (item-invariant): which (row, piece) of the images lane `lane` of this wave copies with its i-th wave-instruction
    unsigned x_rel[Q_NXI]; int x_yx[Q_NXI];            // x_yx = (iy << 8) | ix, or -1
#pragma unroll
    for (int i = 0; i < Q_NXI; ++i) {
        const int k = wave + DMAW * i, R = G::RPP * k + lane / G::PR;
        const int iy = R / Q_INW, ix = R - iy * Q_INW, pc = (lane % G::PR) ^ G::f(ix);
        x_yx[i] = (k < Q_XP && R < Q_NPIX) ? ((iy << 8) | ix) : -1;
        x_rel[i] = 2u * (unsigned)((iy * p.W + ix) * p.ldin + 8 * pc);
    }
    unsigned w_rel[NWI]; int w_co[NWI];                // w_co: cout within the tile, or -1
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int k = wave + DMAW * i, R = G::RPP * k + lane / G::PR, pc = (lane % G::PR) ^ G::f(R);
        const int t = R / BN, co = R - t * BN;
        w_co[i] = k < WP ? co : -1;
        // SPADE: local couts 0..31 are the gamma couts of the workgroup's 32 channels, 32..63 their beta couts (C further on in the filter)
        const int co_g = SPADE ? ((co >> 5) ? p.C : 0) + (co & 31) : co;
        w_rel[i] = 2u * (unsigned)((p.widx[t < 9 ? t : 0] * p.Cout + co_g) * p.Cin + 8 * pc);
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    for (int c = tid; c < Q_BIAS; c += NT) Bs[c] = (p.bias != nullptr && c < p.Cout) ? p.bias[c] : 0.f;
    const unsigned lds_raw = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem_q;

    const int grid = gridDim.x;
    const int u0 = mrdis_xcd_remap(blockIdx.x, grid);
    const int nmine = (p.units - u0 + grid - 1) / grid;           // host: grid <= units
    const int total = nmine * p.nchunks;
    auto decode = [&](int j, int& n, int& a0, int& b0, int& co0) {
        int u = u0 + j * grid;
        const int cot = u % p.coTiles; u /= p.coTiles;
        const int tb = u % p.tilesB; u /= p.tilesB;
        const int ta = u % p.tilesA;
        n = u / p.tilesA; a0 = ta * Q_TH; b0 = tb * Q_TW; co0 = cot * BN;
    };

    // ---- DMA cursor: unit lj, chunk lc
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
    // the item whose DMA is being issued: a snapshot of the cursor (wave-uniform scalars); the piece offsets are formed when a piece is issued (kept in registers
    // they cost 2 x 19 VGPRs in the one-wave-per-SIMD form)
    int c_h0 = 0, c_w0 = 0, c_co0 = 0; unsigned c_xorg = 0, c_c0b = 0; bool c_live = false;
    auto next_offsets = [&]() {
        c_h0 = l_h0; c_w0 = l_w0; c_co0 = l_co0; c_xorg = l_xorg; c_live = l_live; c_c0b = 2u * (unsigned)(lc * Q_KC);
        if (++lc == p.nchunks) { lc = 0; ++lj; load_unit(); }
    };
    auto x_off = [&](int i) -> unsigned {
        const int h = c_h0 + (x_yx[i] >> 8), w_ = c_w0 + (x_yx[i] & 255);
        const bool ok = c_live && x_yx[i] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
        return ok ? c_xorg + x_rel[i] + c_c0b : Q_OOB;
    };
    auto w_off = [&](int i) -> unsigned {
        const bool ok = c_live && w_co[i] >= 0 && (SPADE ? c_co0 / 2 + (w_co[i] & 31) < p.C : c_co0 + w_co[i] < p.Cout);
        return ok ? w_rel[i] + 2u * (unsigned)((SPADE ? c_co0 / 2 : c_co0) * p.Cin) + c_c0b : Q_OOB;
    };
    // one DMA wave-instruction: 64 lanes x 16 bytes land at LDS byte m0v + 16 lane.  Inline assembly: the compiler neither tracks these in vmcnt (the wait
    // before the barrier below is explicit) nor orders LDS reads behind them.
    auto dma = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned lds_byte) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_raw + lds_byte);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(m0v), "s"(rs) : "memory");
    };
    auto dma_x = [&](int stage, int i) { if (wave < DMAW && wave + DMAW * i < Q_XP && !((ABL & 2) && stage >= 0 && lj > 1)) dma(rs_in, x_off(i), (unsigned)(XBASE + stage * Q_XBYTES + 1024 * (wave + DMAW * i))); };
    auto dma_w = [&](int stage, int i) { if (wave < DMAW && wave + DMAW * i < WP && !((ABL & 2) && lj > 1)) dma(rs_w, w_off(i), (unsigned)(stage * WBYTES + 1024 * (wave + DMAW * i))); };

    f32x16 acc[RPW][WC];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int j = 0; j < WC; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[r][j][k] = 0.f;

    if ((p.prio == 1 && wave >= NW / 2) || (p.prio == 2 && wave < NW / 2)) __builtin_amdgcn_s_setprio(1);
    // ---- prologue: item 0 into stage 0
    load_unit();
    next_offsets();
#pragma unroll
    for (int i = 0; i < Q_NXI; ++i) dma_x(0, i);
#pragma unroll
    for (int i = 0; i < NWI; ++i) dma_w(0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int n_stamp = 0;
    auto stamp = [&](int tag) {
        if constexpr ((ABL & 64) != 0) {
            if (p.dbg != nullptr && blockIdx.x < 4 && n_stamp < p.dbg_cap) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) p.dbg[((long long)(blockIdx.x * 8 + wave)) * p.dbg_cap + n_stamp] = (t << 4) | (unsigned)tag;
                ++n_stamp;
            }
        }
    };
    int mj = 0, mc = 0;
    auto iteration = [&](auto P_) {
        constexpr int P = decltype(P_)::value;
        stamp(1);
        const unsigned char* wa0 = smem_q + q_opaque(P * WBYTES + aaddr);
        const unsigned char* wa1 = smem_q + q_opaque(P * WBYTES + (aaddr ^ 32));
        const unsigned char* xs = smem_q + P * Q_XBYTES;                // + xbase[dw] (^ 32) + the tap row: stage and row offsets stay in the instruction's immediate
        next_offsets();                                // item i + 1: its DMA goes out in the first steps below, into stage P ^ 1
        // operand registers in a ring of three: the reads of step s + 2 are issued in step s (a full step of MFMAs -- this wave's and its SIMD partner's --
        // covers their LDS latency; one step ahead left the matrix pipe waiting: 68 us of MFMA + operand reads + barriers against 37 us of MFMAs on 128 -> 256 at 64x64)
        bq_bf16x8 af[3][WC], bf[3][RPW];
        constexpr int NSTEP = 9 * G::KS;
        auto ld_ops = [&](int slot, int s1) {
            const int t1 = s1 / G::KS, ks1 = s1 % G::KS;
#pragma unroll
            for (int j = 0; j < WC; ++j) af[slot][j] = *reinterpret_cast<const bq_bf16x8*>((ks1 ? wa1 : wa0) + (t1 * BN + 32 * j) * RB);
#pragma unroll
            for (int r = 0; r < RPW; ++r) bf[slot][r] = *reinterpret_cast<const bq_bf16x8*>(xs + (xbase[tap_dw(t1) + 1] ^ (ks1 << 5)) + (r + 1 + tap_dh(t1)) * (Q_INW * RB));
        };
        ld_ops(0, 0); ld_ops(1, 1);
        static_assert(Q_NXI <= NSTEP && NWI <= NSTEP, "at most one pixel and one filter piece per step");
        constexpr bool TWO_DMA = Q_NXI + NWI > NSTEP;     // (the 4-wave form: 10 + 9 pieces per wave in 18 steps)
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) {
            const int c_ = s_ % 3;
            if (s_ + 2 < NSTEP) ld_ops((s_ + 2) % 3, s_ + 2);
            if (TWO_DMA) { if (s_ < Q_NXI) dma_x(P ^ 1, s_); if (s_ < NWI) dma_w(P ^ 1, s_); }
            else if (s_ < Q_NXI) dma_x(P ^ 1, s_);
            else if (s_ < Q_NXI + NWI) dma_w(P ^ 1, s_ - Q_NXI);
#pragma unroll
            for (int r = 0; r < RPW; ++r)
#pragma unroll
                for (int j = 0; j < WC; ++j) { if (ABL & 1) acc[r][j][0] += (float)af[c_][j][0] * (float)bf[c_][r][0]; else acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c_][j], bf[c_][r], acc[r][j], 0, 0, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(2);
        if (!(ABL & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of item i + 1 have landed
        stamp(3);
        __syncthreads();
        stamp(4);

        if (++mc == p.nchunks) {
            // ---- epilogue of unit mj: D[cout][position]; a lane owns positions (2 wave + r, e) and the couts 8 g + 4 half .. + 3 (g = 0..3) of each 32-cout block
            stamp(5);
            int n, a0, b0, co0; decode(mj, n, a0, b0, co0);
            mc = 0; ++mj;
            const int b = b0 + e;
            // block-uniform fast path: the whole 16 x 32 tile and all BN couts lie inside, 16-byte stores -- no per-lane predicates, no branches per store (the
            // general path below tests every store: its epilogue took ~4,200 cycles per unit, a third of an item's time on the 256x256 level)
            const bool full = p.wide && a0 + Q_TH <= p.H && b0 + Q_TW <= p.W && (SPADE ? co0 / 2 + 32 <= p.C : co0 + BN <= p.Cout) && !(ABL & 4);
            if (full) {
                const float slope = p.lrelu ? 0.2f : 1.f;
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const long long pix = (long long)(n * p.H + a0 + RPW * wave + r) * p.W + b;
                    if (SPADE) {
                        const int c0 = co0 / 2;
                        const __bf16* zp = reinterpret_cast<const __bf16*>(p.z) + pix * p.ldz + c0 + 4 * half;
                        bq_bf16x4 zq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) zq[q] = *reinterpret_cast<const bq_bf16x4*>(zp + 8 * q);
                        __bf16* mixp = reinterpret_cast<__bf16*>(p.out) + pix * p.ldout + c0;
                        __bf16* gamp = reinterpret_cast<__bf16*>(p.gamma_out) + pix * p.ldg + c0;
                        bq_u32x2 pko[4], pkg[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int ch = c0 + 8 * q + 4 * half;
                            const float4 mu = *reinterpret_cast<const float4*>(p.mean + (long long)n * p.C + ch);
                            const float4 rs = *reinterpret_cast<const float4*>(p.rstd + (long long)n * p.C + ch);
                            const float4 bg = *reinterpret_cast<const float4*>(Bs + ch);
                            const float4 bb = *reinterpret_cast<const float4*>(Bs + p.C + ch);
                            const float g[4] = {acc[r][0][4 * q] + bg.x, acc[r][0][4 * q + 1] + bg.y, acc[r][0][4 * q + 2] + bg.z, acc[r][0][4 * q + 3] + bg.w};
                            const float bt[4] = {acc[r][WC - 1][4 * q] + bb.x, acc[r][WC - 1][4 * q + 1] + bb.y, acc[r][WC - 1][4 * q + 2] + bb.z, acc[r][WC - 1][4 * q + 3] + bb.w};
                            const float m_[4] = {mu.x, mu.y, mu.z, mu.w}, r_[4] = {rs.x, rs.y, rs.z, rs.w};
                            bq_bf16x4 o, og;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float gr = (float)(__bf16)g[k], br = (float)(__bf16)bt[k];
                                og[k] = (__bf16)g[k];
                                o[k] = (__bf16)(((float)zq[q][k] - m_[k]) * r_[k] * (1.f + gr) + br);
                            }
                            pko[q] = __builtin_bit_cast(bq_u32x2, o); pkg[q] = __builtin_bit_cast(bq_u32x2, og);
                        }
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const bq_u32x4 wo_ = bq_pair8(pko[q], pko[q + 1]), wg = bq_pair8(pkg[q], pkg[q + 1]);
                            *reinterpret_cast<bq_u32x4*>(mixp + 8 * (q + half)) = wo_; *reinterpret_cast<bq_u32x4*>(gamp + 8 * (q + half)) = wg;
                        }
                    } else {
                        __bf16* dst = reinterpret_cast<__bf16*>(p.out) + pix * p.ldout + co0;
#pragma unroll
                        for (int j = 0; j < WC; ++j) {
                            bq_u32x2 pk[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 bb = *reinterpret_cast<const float4*>(Bs + co0 + 32 * j + 8 * q + 4 * half);
                                float v[4] = {acc[r][j][4 * q] + bb.x, acc[r][j][4 * q + 1] + bb.y, acc[r][j][4 * q + 2] + bb.z, acc[r][j][4 * q + 3] + bb.w};
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], slope * v[k]);      // LeakyReLU(0.2), or the identity (slope 1): same values as the select
                                bq_bf16x4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
                                pk[q] = __builtin_bit_cast(bq_u32x2, o);
                            }
#pragma unroll
                            for (int q = 0; q < 4; q += 2) *reinterpret_cast<bq_u32x4*>(dst + 32 * j + 8 * (q + half)) = bq_pair8(pk[q], pk[q + 1]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < WC; ++j)
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[r][j][k] = 0.f;
                }
            } else {
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int a = a0 + RPW * wave + r;
                const bool pos_ok = a < p.H && b < p.W && !((ABL & 4) && acc[r][0][0] != 1.2345f);
                const long long pix = (long long)(n * p.H + a) * p.W + b;
                if (SPADE) {
                    const int c0 = co0 / 2;                          // the workgroup's first channel
                    const __bf16* zp = reinterpret_cast<const __bf16*>(p.z) + pix * p.ldz;
                    bq_bf16x4 zq[4]; float4 mu[4], rs[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                  // loads first
                        const int ch = c0 + 8 * q + 4 * half;
                        const bool ok = pos_ok && ch < p.C;
                        zq[q] = ok ? *reinterpret_cast<const bq_bf16x4*>(zp + ch) : bq_bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                        mu[q] = ch < p.C ? *reinterpret_cast<const float4*>(p.mean + (long long)n * p.C + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                        rs[q] = ch < p.C ? *reinterpret_cast<const float4*>(p.rstd + (long long)n * p.C + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    __bf16* mixp = reinterpret_cast<__bf16*>(p.out) + pix * p.ldout;
                    __bf16* gamp = reinterpret_cast<__bf16*>(p.gamma_out) + pix * p.ldg;
                    bq_u32x2 pko[4], pkg[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ch = c0 + 8 * q + 4 * half;
                        const float4 bg = *reinterpret_cast<const float4*>(Bs + (ch < p.C ? ch : 0));
                        const float4 bb = *reinterpret_cast<const float4*>(Bs + (ch < p.C ? p.C + ch : 0));
                        const float g[4] = {acc[r][0][4 * q] + bg.x, acc[r][0][4 * q + 1] + bg.y, acc[r][0][4 * q + 2] + bg.z, acc[r][0][4 * q + 3] + bg.w};
                        const float bt[4] = {acc[r][WC - 1][4 * q] + bb.x, acc[r][WC - 1][4 * q + 1] + bb.y, acc[r][WC - 1][4 * q + 2] + bb.z, acc[r][WC - 1][4 * q + 3] + bb.w};
                        const float m_[4] = {mu[q].x, mu[q].y, mu[q].z, mu[q].w}, r_[4] = {rs[q].x, rs[q].y, rs[q].z, rs[q].w};
                        bq_bf16x4 o, og;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            // gamma and beta reach the modulation kernel of the two-step path as bf16: round them the same way
                            const float gr = (float)(__bf16)g[k], br = (float)(__bf16)bt[k];
                            og[k] = (__bf16)g[k];
                            o[k] = (__bf16)(((float)zq[q][k] - m_[k]) * r_[k] * (1.f + gr) + br);
                        }
                        if (p.wide) { pko[q] = __builtin_bit_cast(bq_u32x2, o); pkg[q] = __builtin_bit_cast(bq_u32x2, og); }
                        else if (pos_ok && ch < p.C) { *reinterpret_cast<bq_bf16x4*>(mixp + ch) = o; *reinterpret_cast<bq_bf16x4*>(gamp + ch) = og; }
                    }
                    if (p.wide) {
#pragma unroll
                        for (int q = 0; q < 4; q += 2) {
                            const bq_u32x4 wo_ = bq_pair8(pko[q], pko[q + 1]), wg = bq_pair8(pkg[q], pkg[q + 1]);
                            const int ch = c0 + 8 * (q + half);
                            if (pos_ok && ch < p.C) { *reinterpret_cast<bq_u32x4*>(mixp + ch) = wo_; *reinterpret_cast<bq_u32x4*>(gamp + ch) = wg; }
                        }
                    }
                } else {
                    __bf16* dst = reinterpret_cast<__bf16*>(p.out) + pix * p.ldout;
#pragma unroll
                    for (int j = 0; j < WC; ++j) {
                        bq_u32x2 pk[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int co = co0 + 32 * j + 8 * q + 4 * half;
                            const float4 bb = *reinterpret_cast<const float4*>(Bs + (co < Q_BIAS - 3 ? co : 0));
                            float v[4] = {acc[r][j][4 * q] + bb.x, acc[r][j][4 * q + 1] + bb.y, acc[r][j][4 * q + 2] + bb.z, acc[r][j][4 * q + 3] + bb.w};
                            if (p.lrelu) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.2f * v[k];
                            }
                            bq_bf16x4 o; o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
                            if (p.wide) pk[q] = __builtin_bit_cast(bq_u32x2, o);
                            else if (pos_ok && co < p.Cout) *reinterpret_cast<bq_bf16x4*>(dst + co) = o;
                        }
                        if (p.wide) {
#pragma unroll
                            for (int q = 0; q < 4; q += 2) {
                                const bq_u32x4 w8 = bq_pair8(pk[q], pk[q + 1]);
                                const int co = co0 + 32 * j + 8 * (q + half);          // this lane's eight consecutive couts
                                if (pos_ok && co < p.Cout) *reinterpret_cast<bq_u32x4*>(dst + co) = w8;
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < WC; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[r][j][k] = 0.f;
            }
            }
            stamp(6);
        }
    };
    for (int i = 0; i < total; i += 2) {
        iteration(QIC<0>{});
        if (i + 1 < total) iteration(QIC<1>{});
    }
}

namespace {
constexpr size_t q_lds(int WC, int NW, int KC, int RPW = 2) {
    return 2 * (size_t)(9 * 32 * WC * 2 * KC) + 2 * (size_t)((((RPW * NW + 2) * Q_INW + 1024 / (2 * KC) - 1) / (1024 / (2 * KC))) * 1024) + sizeof(float) * Q_BIAS;
}
int q_ncu() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)bconv4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<2, false, 0, 8, 32, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<1, false, 0, 8, 32, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<2, false, 0, 8, 32, 2, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<1, false, 0, 8, 32, 2, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<2, false, 0, 8, 32, 2, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)bconv4_kernel<1, false, 0, 8, 32, 2, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            false) return -2;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n_cu;
}
// The kernel is written for NW waves x KC-channel items (QGeo); only 8 x 32 is instantiated.  Measured and dropped (round 5): TWO independent 4-wave workgroups per CU
// on 8 x 32 tiles with 16-channel items (63 KB of LDS each), so that one workgroup's epilogue / DMA wait / barrier would run under the other's MFMAs: 102 vs 84 us
// (128 -> 256 at 64x64), 134 vs 107 us (32 -> 64 at 256x256) -- every workgroup stages its own filter image and an item is half as long (twice the barriers per MFMA);
// and ONE wave per SIMD (4 waves x 4 rows: 128 positions x 64 couts, 8 MFMAs per 6 operand reads, accumulators in AGPRs): 107 vs 78 us / 134 vs 121 us -- a lone wave
// does not keep the matrix pipe busy through its own operand waits, DMA issue and epilogue; s_setprio 1 for either half of the waves, or alternating between the two
// waves of a SIMD step by step: within the noise.
constexpr bool q_two_per_cu(long long) { return false; }
}  // namespace

#ifdef BCONV4_ABLATIONS
static unsigned long long* g_q_dbg = nullptr; static int g_q_dbg_cap = 0;
extern "C" void mrdis_debug_bconv4_stamps(void* buf, int cap_per_wave) { g_q_dbg = (unsigned long long*)buf; g_q_dbg_cap = cap_per_wave; }
#endif
// MRDIS_EUNSUPPORTED: the caller (mrdis_run_bconv) takes bconv3_kernel / bconv_kernel
int mrdis_run_bconv4(const TapConvParams& t, hipStream_t s) {
    if (!mrdis_opt(MRDIS_OPT_BCONV4)) return MRDIS_EUNSUPPORTED;
    if (t.dtype != MRDIS_DT_BF16 || !t.w_bf16 || t.ntaps != 9 || t.is != 1 || t.os != 1 || t.oh0 != 0 || t.ow0 != 0) return MRDIS_EUNSUPPORTED;
    if (t.A != t.Hin || t.B != t.Win || t.Hout != t.Hin || t.Wout != t.Win) return MRDIS_EUNSUPPORTED;
    if (t.Cin % 32 != 0 || t.Cout % 4 != 0 || t.Cout < 16 || t.Cout > Q_BIAS || t.Win < 32 || t.Hin < 8 || t.ldin % 8 != 0 || t.ldout % 4 != 0) return MRDIS_EUNSUPPORTED;
    if ((((uintptr_t)t.in | (uintptr_t)t.w_bf16) & 15) != 0 || ((uintptr_t)t.out & 7) != 0) return MRDIS_EUNSUPPORTED;
    BConv4Params p{};
    int wt = 0;
    bool canon = true, flipped = true;                // the kernel walks the taps in table order with compile-time pixel offsets: the forward's or the data gradient's table
    for (int k = 0; k < 9; ++k) {
        canon = canon && t.dh[k] == k / 3 - 1 && t.dw[k] == k % 3 - 1;
        flipped = flipped && t.dh[k] == 1 - k / 3 && t.dw[k] == 1 - k % 3;
        p.dh[k] = t.dh[k]; p.dw[k] = t.dw[k]; p.widx[k] = t.widx[k];
        if (t.widx[k] < 0) return MRDIS_EUNSUPPORTED;
        if (t.widx[k] + 1 > wt) wt = t.widx[k] + 1;
    }
    if (!canon && !flipped) return MRDIS_EUNSUPPORTED;
    const bool flip = !canon;
    const long long in_b = 2LL * (((long long)t.N * t.Hin * t.Win - 1) * t.ldin + t.Cin), w_b = 2LL * wt * t.Cin * t.Cout;
    if (in_b >= 0xffffffe0LL || w_b >= 0xffffffe0LL) return MRDIS_EUNSUPPORTED;
    p.in = t.in; p.w = t.w_bf16; p.bias = t.bias; p.out = t.out;
    p.N = t.N; p.H = t.Hin; p.W = t.Win; p.Cin = t.Cin; p.ldin = t.ldin; p.Cout = t.Cout; p.ldout = t.ldout;
    p.in_bytes = (unsigned)in_b; p.w_bytes = (unsigned)w_b;
    const int WC = t.Cout > 32 ? 2 : 1, BN = 32 * WC;
    const int n_cu = q_ncu();
    if (n_cu < 0) return n_cu == -1 ? MRDIS_ELAUNCH : MRDIS_EUNSUPPORTED;
    const bool two = q_two_per_cu(0);
    const int TH = two ? 8 : 16, KC = two ? 16 : 32;
    p.tilesA = mrdis_cdiv(t.Hin, TH); p.tilesB = mrdis_cdiv(t.Win, Q_TW); p.coTiles = mrdis_cdiv(t.Cout, BN);
    const long long units = (long long)t.N * p.tilesA * p.tilesB * p.coTiles;
    if (units > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    // the launch has to fill the chip: below one unit per workgroup slot bconv3_kernel / bconv_kernel's smaller tiles spread the layer over more CUs
    const int slots = two ? 2 * n_cu : n_cu;
    if (mrdis_opt(MRDIS_OPT_BCONV4) < 2 && units < slots) return MRDIS_EUNSUPPORTED;
    p.units = (int)units; p.nchunks = t.Cin / KC; p.lrelu = (t.epilogue & MRDIS_EPI_LRELU) ? 1 : 0;
    p.prio = mrdis_opt(MRDIS_OPT_MODE) == 3001 ? 1 : (mrdis_opt(MRDIS_OPT_MODE) == 3002 ? 2 : 0);
    p.wide = (t.Cout % 8 == 0 && t.ldout % 8 == 0 && ((uintptr_t)t.out & 15) == 0 && !mrdis_opt(MRDIS_OPT_NOPACK)) ? 1 : 0;
    const int grid = units < slots ? (int)units : slots;
    mrdis_count(MRDIS_CNT_BCONV4);
#ifdef BCONV4_ABLATIONS
    p.dbg = g_q_dbg; p.dbg_cap = g_q_dbg_cap;
    if (WC == 2) {
        const int abl = (int)mrdis_opt(MRDIS_OPT_MODE);
#define QA(a) if (abl == a) { (void)hipFuncSetAttribute((const void*)bconv4_kernel<2, false, a>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        MRDIS_LAUNCH((bconv4_kernel<2, false, a>), dim3(grid), dim3(512), q_lds(2, 8, 32), s, p); MRDIS_CHECK_LAUNCH(); return MRDIS_OK; }
        QA(1) QA(2) QA(4) QA(8) QA(6) QA(14) QA(64) QA(78)
#undef QA
    }
#endif
#define Q_GO(WC_, NW_, RPW_, FL_) do { if (mrdis_opt(MRDIS_OPT_MODE) == 3003) MRDIS_LAUNCH((bconv4_kernel<WC_, false, 0, NW_, 32, RPW_, FL_, 8>), dim3(grid), dim3(64 * NW_), q_lds(WC_, NW_, 32, RPW_), s, p); \
    else MRDIS_LAUNCH((bconv4_kernel<WC_, false, 0, NW_, 32, RPW_, FL_>), dim3(grid), dim3(64 * NW_), q_lds(WC_, NW_, 32, RPW_), s, p); } while (0)
    if (WC == 2) { if (flip) Q_GO(2, 8, 2, true); else Q_GO(2, 8, 2, false); }
    else { if (flip) Q_GO(1, 8, 2, true); else Q_GO(1, 8, 2, false); }
#undef Q_GO
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// SPADE-fused form (see mrdis_run_bconv3_spade, whose contract this keeps).  MRDIS_EUNSUPPORTED: the caller takes bconv3_kernel<2, 0, true>.
int mrdis_run_bconv4_spade(const void* x, int ldx, const void* w_bf16, const float* bias, const void* z, int ldz, const float* mean, const float* rstd,
                           void* mix, int ldmix, void* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s) {
    if (!mrdis_opt(MRDIS_OPT_BCONV4) || !mrdis_opt(MRDIS_OPT_WINO_PIPE) || Ci % 32 != 0 || C % 4 != 0 || C < 16 || 2 * C > Q_BIAS || W < 32 || H < 8) return MRDIS_EUNSUPPORTED;
    if (ldx % 8 != 0 || ldz % 4 != 0 || ldmix % 4 != 0 || ldg % 4 != 0) return MRDIS_EUNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)w_bf16) | ((uintptr_t)mean) | ((uintptr_t)rstd)) & 15) != 0 || ((((uintptr_t)z) | ((uintptr_t)mix) | ((uintptr_t)gamma)) & 7) != 0) return MRDIS_EUNSUPPORTED;
    BConv4Params p{};
    for (int k = 0; k < 9; ++k) { p.dh[k] = k / 3 - 1; p.dw[k] = k % 3 - 1; p.widx[k] = k; }
    const long long in_b = 2LL * (((long long)N * H * W - 1) * ldx + Ci), w_b = 2LL * 9 * Ci * 2 * C;
    if (in_b >= 0xffffffe0LL || w_b >= 0xffffffe0LL) return MRDIS_EUNSUPPORTED;
    p.in = x; p.w = w_bf16; p.bias = bias; p.out = mix;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = 2 * C; p.ldout = ldmix;
    p.z = z; p.ldz = ldz; p.mean = mean; p.rstd = rstd; p.gamma_out = gamma; p.ldg = ldg; p.C = C;
    p.in_bytes = (unsigned)in_b; p.w_bytes = (unsigned)w_b;
    const int n_cu = q_ncu();
    if (n_cu < 0) return n_cu == -1 ? MRDIS_ELAUNCH : MRDIS_EUNSUPPORTED;
    const bool two = q_two_per_cu(0);
    const int TH = two ? 8 : 16, KC = two ? 16 : 32;
    p.tilesA = mrdis_cdiv(H, TH); p.tilesB = mrdis_cdiv(W, Q_TW); p.coTiles = mrdis_cdiv(C, 32);
    const long long units = (long long)N * p.tilesA * p.tilesB * p.coTiles;
    if (units > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    const int slots = two ? 2 * n_cu : n_cu;
    if (mrdis_opt(MRDIS_OPT_BCONV4) < 2 && units < slots) return MRDIS_EUNSUPPORTED;
    p.units = (int)units; p.nchunks = Ci / KC; p.lrelu = 0;
    p.prio = mrdis_opt(MRDIS_OPT_MODE) == 3001 ? 1 : (mrdis_opt(MRDIS_OPT_MODE) == 3002 ? 2 : 0);
    p.wide = (C % 8 == 0 && ldmix % 8 == 0 && ldg % 8 == 0 && ((((uintptr_t)mix) | ((uintptr_t)gamma)) & 15) == 0 && !mrdis_opt(MRDIS_OPT_NOPACK)) ? 1 : 0;
    const int grid = units < slots ? (int)units : slots;
    mrdis_count(MRDIS_CNT_BCONV4_SPADE);
    MRDIS_LAUNCH((bconv4_kernel<2, true>), dim3(grid), dim3(512), q_lds(2, 8, 32), s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
