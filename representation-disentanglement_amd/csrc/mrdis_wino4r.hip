// mrdis_wino4r.hip -- Winograd F(4x4, 3x3) for the layers with <= 32 couts, register-fed form (gfx950, fp32, NHWC): forward / data gradient of
// the 3x3 / stride 1 / pad 1 layers whose filter image has format 5 (reference: F.conv2d inside CondConv2d._conv_forward, model.py:2104-2117).
//
// The MFMA `v_mfma_f32_16x16x4_f32` takes its B operand as lane (n = lane % 16, k = lane / 16) -> B[k][n].  With n = tile and k = channel of
// the 4-channel chunk that is exactly the layout of an input transform done one (tile, channel) per lane: a wave that transforms the 6 x 6
// patches of its 16 tiles x 4 channels holds, point by point, the B operand of its own MFMAs in registers.  So here every wave transforms for
// itself -- 144 VALU + 36 LDS reads per 36 MFMAs -- and the transformed input never goes through LDS: no V buffer, no producer / consumer
// hand-off between waves, one barrier per 8-channel stage.  (mrdis_wino4.hip's narrow form shares one transform between the two 16-cout
// waves of a tile group through LDS and ran at 31 % of the matrix pipe: 64 tiles per 32 couts double its transform work per MFMA.)
//
// Workgroup = 8 waves, wave = 16 tiles (4 x 4 tiles = 16 x 16 outputs) x 16 couts x 36 points = 144 accumulators (two waves per SIMD).
//   NKH = 2: 2 tile groups x 2 cout halves x 2 CHANNEL halves -- 32 tiles (16 x 32 outputs); the wave pair of a (tile group, cout half)
//            splits every 8-channel stage (chunk kh each) and adds its two partial OUTPUTS (after A^T . A: 16 values instead of 36) through
//            LDS at the end.  Half the pixels per workgroup: the 128-byte lines a stage touches (16 or 32 bytes of each) stay in the XCD's
//            L2 until the stages that need the rest of them (32 workgroups x 18 x 34 lines = 2.5 MB of 4 MB; with 64 tiles 4.7 MB: every
//            line re-fetched from the Infinity Cache by all its 8 chunks -- 64 -> 32 at 256x256 then runs at the Infinity Cache's rate).
//   NKH = 1: 4 tile groups x 2 cout halves, 64 tiles (32 x 32 outputs), each wave both chunks of a stage: no exchange, half the filter traffic.
// Stage = 8 channels: the raw block [rows][40 pixels][2 quads][4 channels] by `buffer_load_dwordx4 ... lds` (a lane outside
// the image reads beyond the descriptor: zeros = the padding) + two 18 KB filter chunks by `global_load_lds_dwordx4`, double-buffered; the next
// block's first stage is in flight during a block's epilogue.
// Raw block: pixel (y, x) has index 40 y + ((y >> 2) & 3) + x (the skew keeps the 16 tiles of a wave -- origins 4 ty, 4 tx -- on different banks; see r_b).
#include "mrdis_common.h"
#include "mrdis_wino4.h"

struct Wino4rParams {
    const float* in; const float* bias; float* out; const float* u_img;
    int N, H, W, Cin, ldin, Cout, ldout;
    int lrelu, nt_out;
    int nby, nbx, coTiles, nblk;
    unsigned in_bytes;
    unsigned long long* dbg; int dbg_cap;      // diagnostic build: s_memtime stamps per (workgroup, wave)
};

namespace {
constexpr int NTR = 512;
constexpr int R_RWP = 40;                       // pixel slots per raw row (34 + skew 3, rounded to a multiple of 4)
constexpr int R_UCH = MRDIS_W4N_UCHUNK;         // floats per (32-cout tile, chunk) of the format-5 image
constexpr unsigned R_OOB = 0xfffffff0u;
typedef float f32x2_r __attribute__((ext_vector_type(2)));
template <int V_> struct ICR { static constexpr int value = V_; };
template <int NKH> struct RGeo {
    static constexpr int TGY = NKH == 2 ? 1 : 2;            // tile groups down the block (two across)
    static constexpr int BH = 16 * TGY;                     // output rows per block
    static constexpr int RBH = BH + 2;                      // raw rows
    static constexpr int RSLOTS = RBH * R_RWP;              // 16-byte slots per 4-channel group
    static constexpr int PPS = (2 * RSLOTS + 63) / 64;      // 1-KiB raw copy pieces per stage (the last one moved back to end with the block)
    static constexpr int NRP = (PPS + 7) / 8;               // raw pieces per wave and stage (piece wave + 8 i, clamped to the last: a duplicate copy)
    static constexpr int NP = 5 + NRP;                      // + five of the 36 filter pieces
    static constexpr int NCH = NKH == 2 ? 1 : 2;            // chunks per wave and stage
    static constexpr int RGF = RSLOTS * 4;                  // floats per group
    static constexpr int SBUF = 2 * R_UCH + 2 * RGF;        // floats per stage buffer: two filter chunks, two raw groups
    static constexpr int EXTRA = NKH == 2 ? 16384 - SBUF : 0;      // the output exchange takes 64 KB: [buffer 0][extra][buffer 1], buffer + extra = 64 KB
    static constexpr int BSTRIDE = SBUF + EXTRA;
    static constexpr size_t LDS = sizeof(float) * (2 * SBUF + EXTRA);
};
__device__ __forceinline__ f32x2_r r_ld2(const float* p) { return *(const volatile __attribute__((address_space(3))) f32x2_r*)p; }
// (volatile: one ds_read_b32 each -- hipcc otherwise pairs neighbours into ds_read2_b32, which the LDS serves by the 32-bank rule: two tiles per bank here)
__device__ __forceinline__ float r_ld1(const float* p) { return *(const volatile __attribute__((address_space(3))) float*)p; }
}  // namespace

template <int NKH, int ABL>
__global__ __launch_bounds__(512, 2) void wino4r_kernel(const Wino4rParams p) {
    using G = RGeo<NKH>;
    static_assert(NKH == 1 || G::EXTRA >= 0, "exchange area");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = wave & 1, tgx = (wave >> 1) & 1, w3 = wave >> 2;
    const int kh = NKH == 2 ? w3 : 0, tgy = NKH == 2 ? 0 : w3;
    // MFMA roles: A = U rows (couts 16 cq + l16), B = this lane's transformed patch (tile l16 of the group, channel kq of the chunk)
    const int a_off = kq * 64 + ((2 * (16 * cq + l16) + 32 * kq) & 63);
    const int ty = l16 >> 2, tx = l16 & 3;
    const int ry0 = 16 * tgy + 4 * ty, rx0 = 16 * tgx + 4 * tx;
    // raw block (8 channels of a stage): pixel (y, x) has index pidx = 40 y + ((y >> 2) & 3) + x and its two 4-channel quads sit in the neighbouring 16-byte slots
    // 2 pidx + (q ^ ((x >> 3) & 1)) -- two lanes of a copy read the 32 contiguous bytes of a pixel (one fill of its 128-byte line instead of two: the copies
    // are bound by the L1's line fills, a lane per line), and for a fixed patch position and quad the 16 tiles of a wave still sit at 16 different slots mod 16
    // (2 (skew + 4 (tx & 1)) + quad ^ (tx >> 1 | (tx + 1) >> 1 for patch columns 4, 5)): 64 distinct banks per ds_read_b32.  Four bases: patch rows 0-3 / 4-5
    // (the next tile-row group's skew) x patch columns 0-3 / 4-5; the quad of this wave's chunk is folded in (channel-split form) or XORed in per chunk.
    int r_b[2][2];
#pragma unroll
    for (int hi = 0; hi < 2; ++hi)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
            r_b[hi][ch] = (8 * (ry0 * R_RWP + (((ry0 >> 2) + hi) & 3) + rx0) + kq + 4 * (((tx + ch) >> 1) & 1)) ^ (NKH == 2 ? 4 * kh : 0);

    const int grid = gridDim.x;
    const int rb0 = mrdis_xcd_remap(blockIdx.x, grid);
    const int nmine = (p.nblk - rb0 + grid - 1) / grid;            // host: grid <= nblk
    const int nst = p.Cin >> 3;                                    // host: Cin % 8 == 0
    const int total = nmine * nst;
    auto decode = [&](int j, int& n, int& oy0, int& ox0, int& cot) {
        int b = rb0 + j * grid;
        cot = b % p.coTiles; b /= p.coTiles;
        const int bx = b % p.nbx; b /= p.nbx;
        const int by = b % p.nby;
        n = b / p.nby; oy0 = G::BH * by; ox0 = 32 * bx;
    };

    // ---- staging: a stage's copies are 36 filter pieces + PPS raw pieces of 1 KiB (64 lanes x 16 bytes, landing at M0 + 16 lane); wave w issues filter
    //      pieces w + 8 k and raw pieces w + 8 i, one or two per row step of its chunk(s) so that they queue behind the MFMAs instead of in front of them
    //      (issued in one burst at the top of the stage they held every wave for 1000-2000 cycles: the texture addresser takes 16 cycles per piece).
    //      An index past the last piece repeats the last piece (same bytes to the same place) rather than branching around the copy.
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    int s_yx[G::NRP]; unsigned s_off[G::NRP];
    auto raw_piece = [&](int i, int& start) {
        int r = wave + 8 * i;
        if (r > G::PPS - 1) r = G::PPS - 1;
        start = 64 * r;
        if (start + 64 > 2 * G::RSLOTS) start = 2 * G::RSLOTS - 64;
    };
#pragma unroll
    for (int i = 0; i < G::NRP; ++i) {
        int start; raw_piece(i, start);
        const int slot = start + lane, pidx = slot >> 1, y = pidx / R_RWP, xs = pidx - y * R_RWP - ((y >> 2) & 3);
        const int q = (slot & 1) ^ ((xs >> 3) & 1);
        s_yx[i] = (xs >= 0 && xs < 34) ? ((q << 16) | (y << 8) | xs) : -1;
        s_off[i] = R_OOB;
    }
    int dj = 0, ds = 0;                                            // staging cursor: block index, stage
    const float* f_blk = p.u_img;                                  // filter image of the cursor's cout tile
    auto stage_block = [&]() {
#pragma unroll
        for (int i = 0; i < G::NRP; ++i) s_off[i] = R_OOB;
        f_blk = p.u_img;
        if (dj < nmine) {
            int n, oy0, ox0, cot; decode(dj, n, oy0, ox0, cot);
            f_blk = p.u_img + (long long)cot * (2 * nst) * R_UCH;
#pragma unroll
            for (int i = 0; i < G::NRP; ++i) {
                const int h = oy0 - 1 + ((s_yx[i] >> 8) & 255), w_ = ox0 - 1 + (s_yx[i] & 255);
                if (s_yx[i] >= 0 && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                    s_off[i] = 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldin) + 16u * (unsigned)(s_yx[i] >> 16);          // host: < 2^30 elements
            }
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned f_voff = 16u * (unsigned)lane;
    auto stage_piece = [&](int buf, int i) {                       // copy piece i (compile-time) of this wave for the cursor's stage into buffer `buf`
        const unsigned base = lds0 + 4u * (unsigned)(buf * G::BSTRIDE);
        if (i < 5) {
            if (ABL & 8) return;
            int q = wave + 8 * i;
            if (q > 35) q = 35;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(base + 1024u * (unsigned)q);
            const float* src = f_blk + (long long)ds * (2 * R_UCH) + 256 * q;
            unsigned keep;                                         // M0 is the compiler's: written and restored inside the one statement that reads it
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(m0v), "v"(f_voff), "s"(src) : "memory");
        } else {
            if (ABL & 32) return;
            int start; raw_piece(i - 5, start);
            const unsigned soff = __builtin_amdgcn_readfirstlane(32u * (unsigned)ds);
            const unsigned m0v = __builtin_amdgcn_readfirstlane(base + 4u * (unsigned)(2 * R_UCH) + 16u * (unsigned)start);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(s_off[i - 5]), "s"(m0v), "s"(rs_in), "s"(soff) : "memory");
        }
    };
    auto stage_advance = [&]() { if (++ds == nst) { ds = 0; ++dj; stage_block(); } };

    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
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

    // ---- one 4-channel chunk: patch -> B^T d B (B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]) -> 36 MFMAs.
    //      Software-pipelined by hand (hipcc otherwise sinks every filter read to just above its MFMAs: one exposed LDS latency per pair): step a_ = the six
    //      MFMAs of row a_ + row a_ + 1 of B^T d (only the column arithmetic that row needs: 12 / 18 / 6 / 18 / 6 / 12 VALU for rows 0-5, so the MFMAs start
    //      24 VALU after the patch has landed instead of 84) and its row transform (12 VALU) + the three filter reads of row a_ + 3 + this wave's share of
    //      the next stage's copies, fenced so nothing leaves its step.  CI: index of the chunk in the stage (which copy pieces its steps issue into buffer nbuf).
    float d[36];                                                   // the patch: d[6 r + c] = pixel (r, c) of this lane's (tile, channel)
    auto patch_rows = [&](const float* Rg, int q, int par) {       // rows par, par + 2, par + 4 of quad q
#pragma unroll
        for (int r = par; r < 6; r += 2)
#pragma unroll
            for (int c = 0; c < 6; ++c) d[6 * r + c] = (ABL & 1) ? 1.f : r_ld1(Rg + (r_b[r >= 4][c >= 4] ^ (q ? 4 : 0)) + 8 * (r * R_RWP + c));
    };
    auto chunk = [&](const float* Uc, const float* Rg, auto CI_, int nbuf) {
        constexpr int CI = decltype(CI_)::value;
        float ca[6], cb[6], tr[6], v[2][6];
        f32x2_r u[4][3];
        const float* Ua = Uc + a_off;
        // 64-tile form: the second chunk's patch is read during the first chunk's MFMAs (rows 0, 2, 4 in step 3, rows 1, 3, 5 in step 5: into the registers
        // the first patch has vacated by then), so only one patch read per stage stands in front of the MFMAs
        if (NKH == 2 || CI == 0) { patch_rows(Rg, 0, 0); patch_rows(Rg, 0, 1); }
#pragma unroll
        for (int a_ = 0; a_ < 3; ++a_)
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) u[a_][b2] = (ABL & 2) ? f32x2_r{1.f, 1.f} : r_ld2(Ua + (3 * a_ + b2) * 256);
        auto col_row = [&](int a_) {                               // row a_ of B^T d into tr[]
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float e0 = d[c], e1 = d[6 + c], e2 = d[12 + c], e3 = d[18 + c], e4 = d[24 + c], e5 = d[30 + c];
                if (a_ == 0) tr[c] = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
                else if (a_ == 1) { ca[c] = fmaf(-4.f, e2, e4); cb[c] = fmaf(-4.f, e1, e3); tr[c] = ca[c] + cb[c]; }
                else if (a_ == 2) tr[c] = ca[c] - cb[c];
                else if (a_ == 3) { ca[c] = e4 - e2; cb[c] = e3 - e1; tr[c] = fmaf(2.f, cb[c], ca[c]); }
                else if (a_ == 4) tr[c] = fmaf(-2.f, cb[c], ca[c]);
                else tr[c] = fmaf(4.f, e1, fmaf(-5.f, e3, e5));
            }
        };
        auto row_op = [&](float* vo) {                             // (B^T d) B for the row in tr[]
            const float r0 = tr[0], r1 = tr[1], r2 = tr[2], r3 = tr[3], r4 = tr[4], r5 = tr[5];
            const float aa = fmaf(-4.f, r2, r4), bb = fmaf(-4.f, r1, r3), cc = r4 - r2, ee = r3 - r1;
            vo[0] = fmaf(4.f, r0, fmaf(-5.f, r2, r4)); vo[1] = aa + bb; vo[2] = aa - bb;
            vo[3] = fmaf(2.f, ee, cc); vo[4] = fmaf(-2.f, ee, cc); vo[5] = fmaf(4.f, r1, fmaf(-5.f, r3, r5));
        };
        col_row(0); row_op(v[0]);
        __builtin_amdgcn_sched_barrier(0);
        stamp(6);
#pragma unroll
        for (int a_ = 0; a_ < 6; ++a_) {
            constexpr int TS = 6 * G::NCH;
            const int gsi = 6 * CI + a_;
            const int p0 = gsi * G::NP / TS, p1 = (gsi + 1) * G::NP / TS;
#pragma unroll
            for (int i = 0; i < G::NP; ++i) if (i >= p0 && i < p1) stage_piece(nbuf, i);
            if (a_ + 3 < 6) {
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) u[(a_ + 3) & 3][b2] = (ABL & 2) ? f32x2_r{1.f, 1.f} : r_ld2(Ua + (3 * (a_ + 3) + b2) * 256);
            }
            if (a_ + 1 < 6) { col_row(a_ + 1); row_op(v[(a_ + 1) & 1]); }
            if (NKH == 1 && CI == 0 && a_ == 3) patch_rows(Rg, 1, 0);
            if (NKH == 1 && CI == 0 && a_ == 5) patch_rows(Rg, 1, 1);
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) {
                const int pp = 3 * a_ + b2;
                const f32x2_r uu = u[a_ & 3][b2];
                if (!(ABL & 4)) {
                    acc[2 * pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(uu.x, v[a_ & 1][2 * b2], acc[2 * pp], 0, 0, 0);
                    acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uu.y, v[a_ & 1][2 * b2 + 1], acc[2 * pp + 1], 0, 0, 0);
                } else { acc[2 * pp][0] += uu.x * v[a_ & 1][2 * b2]; acc[2 * pp + 1][0] += uu.y * v[a_ & 1][2 * b2 + 1]; }
            }
#pragma unroll
            for (int g_ = 0; g_ < 6; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA,
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // at most one copy,
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);      // up to three LDS reads
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // and five VALU in its shadow
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);                  // the second-dispatched half loses every arbitration on its SIMD otherwise (MI355X_MICROARCH.md, two waves per SIMD)
    stage_block();
#pragma unroll
    for (int i = 0; i < G::NP; ++i) stage_piece(0, i);
    stage_advance();
    int cj = 0, cs = 0;                                            // compute cursor
    for (int gs = 0; gs < total; ++gs) {
        const int buf = gs & 1;
        stamp(1);
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): this wave's copies of stage gs (and the last epilogue's stores)
        __syncthreads();                                           // every wave's copies have landed; every wave is done with buffer buf ^ 1
        stamp(2);
        const int nbuf = buf ^ 1;                                  // (past the last stage the cursor's copies read zeros / filter chunk 0 into a buffer nobody reads)
        const float* Sb = smem + buf * G::BSTRIDE;
        if constexpr (NKH == 2) {
            chunk(Sb + kh * R_UCH, Sb + 2 * R_UCH, ICR<0>{}, nbuf);
        } else {
            chunk(Sb, Sb + 2 * R_UCH, ICR<0>{}, nbuf);
            chunk(Sb + R_UCH, Sb + 2 * R_UCH, ICR<1>{}, nbuf);
        }
        stage_advance();
        stamp(3);
        if (++cs != nst) continue;
        cs = 0;
        // ---- epilogue of block cj: lane = tile l16 of group (tgy, tgx), couts 32 cot + 16 cq + 4 kq + r; Y = A^T M A,
        //      A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
        int n, oy0, ox0, cot; decode(cj, n, oy0, ox0, cot);
        ++cj;
        const int oy = oy0 + ry0, ox = ox0 + rx0;
        const int co = 32 * cot + 16 * cq + 4 * kq;
        const bool co_ok = co < p.Cout;                            // host: Cout % 4 == 0
        const bool full = oy0 + G::BH <= p.H && ox0 + 32 <= p.W && 32 * cot + 32 <= p.Cout;      // block-uniform: no per-store tests
        const f32x4 slope = p.lrelu ? f32x4{0.2f, 0.2f, 0.2f, 0.2f} : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 b4 = (p.bias != nullptr && co_ok) ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        float* const o00 = p.out + ((long long)(n * p.H + oy) * p.W + ox) * p.ldout + co;
        const long long rowp = (long long)p.W * p.ldout;
        auto out_row = [&](int i, f32x4* y) {                      // output row i of the tile: four pixels x four couts
            f32x4 t[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const f32x4 m1 = acc[6 + b], m2 = acc[12 + b], m3 = acc[18 + b], m4 = acc[24 + b];
                if (i == 0) t[b] = acc[b] + (m1 + m2) + (m3 + m4);
                else if (i == 1) t[b] = (m1 - m2) + 2.f * (m3 - m4);
                else if (i == 2) t[b] = (m1 + m2) + 4.f * (m3 + m4);
                else t[b] = (m1 - m2) + 8.f * (m3 - m4) + acc[30 + b];
            }
            const f32x4 s12 = t[1] + t[2], d12 = t[1] - t[2], s34 = t[3] + t[4], d34 = t[3] - t[4];
            y[0] = t[0] + s12 + s34; y[1] = d12 + 2.f * d34; y[2] = s12 + 4.f * s34; y[3] = d12 + 8.f * d34 + t[5];
        };
        auto put = [&](int i, int k, f32x4 v) {                    // (the bias is added by the caller, before the first conditional store: a wait for its load inside
            v = __builtin_elementwise_max(v, v * slope);           //  the store branches would be vmcnt(0) -- the stores too -- once per store)
            float* dst = o00 + i * rowp + k * p.ldout;
            if (full || (co_ok && oy + i < p.H && ox + k < p.W)) {
                if (p.nt_out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                else *reinterpret_cast<f32x4*>(dst) = v;
            }
        };
        if constexpr (NKH == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 y[4];
                out_row(i, y);
#pragma unroll
                for (int k = 0; k < 4; ++k) put(i, k, y[k] + b4);
            }
        } else {
            // the wave pair (kh = 0, 1) of a (tile group, cout half) holds two partial sums of the same outputs: wave kh keeps output rows 2 kh, 2 kh + 1 of
            // every tile and hands the other two to its partner through the stage buffer the block has just finished with
            float* const xch = smem + (buf ? G::SBUF : 0);         // 64 KB: buffer 0 + extra | extra + buffer 1
            f32x4 mine[8];
            __syncthreads();                                       // every wave is done reading buffer `buf`
#pragma unroll
            for (int e = 0; e < 2; ++e) {                          // branch-free: a wave-uniform branch here would put the 144 accumulators through a phi
                f32x4 ya[4], yb[4];
                out_row(e, ya); out_row(e + 2, yb);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 keep_, send_;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { keep_[c] = kh ? yb[k][c] : ya[k][c]; send_[c] = kh ? ya[k][c] : yb[k][c]; }
                    mine[4 * e + k] = keep_ + b4;
                    *reinterpret_cast<f32x4*>(xch + wave * 2048 + (4 * e + k) * 256 + 4 * lane) = send_;
                }
            }
            __syncthreads();
            const float* theirs = xch + (wave ^ 4) * 2048 + 4 * lane;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(theirs + e * 256);
                put(2 * kh + (e >> 2), e & 3, mine[e] + o);
            }
        }
#pragma unroll
        for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        stamp(4);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // the copies issued past the last stage land before the wave ends
}

#ifdef WINO4_ABLATIONS
static unsigned long long* g_w4r_dbg = nullptr; static int g_w4r_dbg_cap = 0;
extern "C" void mrdis_debug_wino4r_stamps(void* buf, int cap_per_wave) { g_w4r_dbg = (unsigned long long*)buf; g_w4r_dbg_cap = cap_per_wave; }
#endif

template <int NKH>
static int launch_wino4r(Wino4rParams& p, hipStream_t s) {
    using G = RGeo<NKH>;
    p.nby = mrdis_cdiv(p.H, G::BH); p.nbx = mrdis_cdiv(p.W, 32);
    p.coTiles = mrdis_cdiv(p.Cout, 32);
    const long long nblk = (long long)p.N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (mrdis_opt(MRDIS_OPT_WINO4) < 2 && nblk < 192) return MRDIS_EUNSUPPORTED;
    p.nblk = (int)nblk;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)wino4r_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RGeo<1>::LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)wino4r_kernel<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RGeo<2>::LDS) != hipSuccess)
            return MRDIS_EUNSUPPORTED;
#ifdef WINO4_ABLATIONS
#define W4RA(a) hipFuncSetAttribute((const void*)wino4r_kernel<1, a>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RGeo<1>::LDS); \
                hipFuncSetAttribute((const void*)wino4r_kernel<2, a>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RGeo<2>::LDS);
        W4RA(64) W4RA(1) W4RA(2) W4RA(3) W4RA(4) W4RA(40) W4RA(43) W4RA(47)
#undef W4RA
#endif
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = nblk < n_cu ? (int)nblk : n_cu;
#ifdef WINO4_ABLATIONS
    {   // timing-only variants (results wrong), selected by option debug_mode: 1 no patch reads, 2 no filter reads, 4 no MFMAs, 8 | 32 no copies
        const int abl = (int)mrdis_opt(MRDIS_OPT_MODE);
#define W4RA(a) if (abl == a) { MRDIS_LAUNCH((wino4r_kernel<NKH, a>), dim3(grid), dim3(NTR), G::LDS, s, p); MRDIS_CHECK_LAUNCH(); return MRDIS_OK; }
        W4RA(1) W4RA(2) W4RA(3) W4RA(4) W4RA(40) W4RA(43) W4RA(47)
#undef W4RA
    }
    if (g_w4r_dbg != nullptr) {
        p.dbg = g_w4r_dbg; p.dbg_cap = g_w4r_dbg_cap;
        mrdis_count(MRDIS_CNT_WINO4R);
        MRDIS_LAUNCH((wino4r_kernel<NKH, 64>), dim3(grid), dim3(NTR), G::LDS, s, p);
        MRDIS_CHECK_LAUNCH();
        return MRDIS_OK;
    }
#endif
    mrdis_count(MRDIS_CNT_WINO4R);
    MRDIS_LAUNCH((wino4r_kernel<NKH, 0>), dim3(grid), dim3(NTR), G::LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// u_img: the format-5 image of the layer's filter (mrdis_wino2.hip builds it).  Option wino4r: 0 = never (the shared-transform form of mrdis_wino4.hip where that
// applies), 1 = the 64-tile form for <= 64 reduction channels and for inputs beyond the Infinity Cache, which the shared-transform form declines (B = 32: 64 -> 32
// at 256x256 346 us against 476 for F(2x2) and 497 for the shared-transform form; at 128x128 88 against 101; 128 -> 32 at 128x128 level, 16 -> 32 behind), 2 / 3 =
// always the 64-tile / the channel-split form
int mrdis_run_wino4r(const float* x, int ldx, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu,
                     hipStream_t s, const float* u_img) {
    const long long mode = mrdis_opt(MRDIS_OPT_WINO4R);
    if (mode == 0 || !u_img || (((uintptr_t)u_img) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (Ci % 8 != 0 || Ci < 16 || Co < 4 || Co % 4 != 0 || Co > 32 || ldx % 4 != 0 || ldy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (bias != nullptr && (((uintptr_t)bias) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x3fffffffLL || H < 16 || W < 32) return MRDIS_EUNSUPPORTED;
    Wino4rParams p{};
    p.in_bytes = (unsigned)(4LL * ((long long)(N * H) * W - 1) * ldx + 4LL * Ci);
    p.in = x; p.bias = bias; p.out = y; p.u_img = u_img;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldy;
    p.lrelu = lrelu;
    { const long long mb = mrdis_opt(MRDIS_OPT_NT_MB); p.nt_out = (long long)N * H * W * ldy * 4 >= mb * 1000000LL ? 1 : 0; }
    if (mode == 1 && Ci > 64 && (long long)N * H * W * Ci * 4 <= 300000000LL) return MRDIS_EUNSUPPORTED;
    return mode == 3 ? launch_wino4r<2>(p, s) : launch_wino4r<1>(p, s);
}
