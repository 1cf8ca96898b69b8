// mrdis_wino4.hip -- fused Winograd F(4x4, 3x3) convolution for gfx950 (MI355X), fp32, NHWC: forward and data gradient of the
// 3x3 / stride 1 / pad 1 layers with >= 64 couts (reference: F.conv2d inside CondConv2d._conv_forward, model.py:2104-2117).
//
// F(2x2, 3x3) (mrdis_wino2.hip) executes 16 multiplies per 4 outputs, F(4x4, 3x3) 36 per 16: 1.78x fewer MFMAs for the same
// layer, at transforms that cost about the same per output (the 6x6 input transform is 12 FMAs per 6 values, the output transform
// 10 per 4).  Y = A^T [(G g G^T) .* (B^T d B)] A with the interpolation points 0, +-1, +-2, inf; everything is fp32 (exact-fp32 MFMA
// `v_mfma_f32_16x16x4_f32`); the transforms multiply by 2, 4, 5, 8 and (filter side, once per step) 1/4, 1/6, 1/12, 1/24, which moves the
// result by ~5e-6 of its maximum against 1e-6 for the direct kernel (tests/test_gpu_wino4.py; the parity bar is 1e-3).
//
// Workgroup = 8 waves = 32 tiles (4 x 8 tiles = 16 x 32 outputs) x 64 couts; wave = 16 tiles x 16 couts for ALL 36 points: 144
// accumulator registers (two waves per SIMD leave 256).  One iteration = one 4-channel chunk = ONE k-step: 36 MFMAs per wave, the A
// operand (two points of U) and the B operand (two points of V) one ds_read_b64 each per two MFMAs.  Pipeline, one barrier per iteration:
//     iteration i:   36 MFMAs on U(i), V(i)                                                          all waves
//                    V(i+1) = B^T d B from the raw block in LDS (72 VALU + 15 reads + 9 writes per thread)   waves 0-3 ("T")
//                    U(i+1): nine 1-KiB LDS-DMA pieces per wave straight from the filter image (no VGPRs)    waves 4-7 ("S")
//                    raw block of the next 8-channel double chunk: global loads (odd i) / LDS stores (even i)  waves 4-7
// A SIMD hosts one T and one S wave, so the VALU-heavy transform and the memory-side staging share its issue slots.
// LDS: U 2 x 36 KB + V 2 x 18 KB + raw 2 x 22.5 KB + bias 2 KB = 155 KB, one persistent workgroup per CU walking (block, chunk)
// pairs, so a block's prologue hides under its predecessor's MFMAs.
//
// LDS images (conflict-free without padding):
//   U / V point pair pp (points 2 pp, 2 pp + 1): [4 channels kq][128 | 64 slots]; slot of (cout | tile m, parity) =
//     (2 m + parity + 32 kq) & (127 | 63): a half-wave's b64 operand reads (16 m x 2 kq) cover 64 distinct banks.
//   raw block: channel planes [8][18 rows][40] with rows of tile-row group g = row / 4 shifted by 2 * ((g >> 1) & 1) floats: the
//     b64 reads of a transform wave (8 tile columns x 4 tile rows, row i of each tile) then hit {0, 32, 2, 34} + 4 tx: 64 distinct banks.
#include "mrdis_common.h"
#include "mrdis_wino4.h"

struct Wino4Params {
    const float* in; const float* bias; float* out; const float* u_img;
    int N, H, W, Cin, ldin, Cout, ldout;
    int lrelu, nt_out, prio;
    int nby, nbx, coTiles, nblk;      // 16 x 32-output blocks per image, 64-cout tiles, blocks in total
    unsigned in_bytes;                // record count of the input's buffer descriptor
    // SPADE epilogue (wino4_kernel<., true>): the image is that of the fused gamma | beta filter of a SPADE block (Cout = 2 C) in the cout order
    // [32-channel tile][8-channel group cg][gamma of the 8 channels | beta of the 8 channels]: the lower half-wave of an accumulator holds gamma, the
    // upper half-wave beta of the same four channels; one v_permlane32_swap per value pairs them up, and the epilogue writes
    // out = (z - mean) * rstd * (1 + gamma) + beta and gamma itself (model.py:2440-2446) -- the 2C-channel tensor never exists
    const float* z; const float* mean; const float* rstd; float* gamma_out;
    int ldz, ldg, C;
    unsigned z_bytes;                 // record count of z's buffer descriptor (0: no prefetch of z, see the block loop)
    unsigned long long* dbg; int dbg_cap;    // diagnostic build (ABL & 64): per (workgroup, wave) s_memtime stamps, dbg_cap per wave
};

namespace {
constexpr int NT4 = 512;
constexpr int KC4 = MRDIS_W4_KC;
constexpr int UPP = MRDIS_W4_UPP, UBUF = MRDIS_W4_UCHUNK;      // 512, 9216 floats
constexpr int VPP = 256, VBUF = 18 * VPP;                      // 4608 floats
constexpr int RWP = 40, PL = 18 * RWP, RAWBUF = 8 * PL;        // 720, 5760 floats
constexpr int RBW = 34, RBH = 18, NITEM = RBW * RBH * 2;       // raw block: 18 x 34 pixels x two 4-channel quads
constexpr int NIT = (NITEM + 255) / 256;                       // staging items per S thread (5)
constexpr int BIAS4 = 512;
constexpr int W4_SINK = 8 * 64;                                // floats: a 256-byte landing strip per wave for the z prefetch (SPADE form; never read)
constexpr size_t W4_LDS = sizeof(float) * (2 * UBUF + 2 * VBUF + 2 * RAWBUF + BIAS4 + W4_SINK);
template <int V_> struct IC4 { static constexpr int value = V_; };
typedef unsigned u32x4_w4 __attribute__((ext_vector_type(4)));
typedef float f32x2_w4 __attribute__((ext_vector_type(2)));
// 8-byte LDS accesses that stay 8-byte accesses: volatile keeps hipcc from pairing neighbours into ds_read2_b64 / ds_write2st64_b64 (half the
// rate of two ds_read_b64, banked by 32 where these images are laid out for the 64-bank rule of the 8-byte forms: 0.5 conflict cycles per LDS
// instruction measured); the explicit LDS address space keeps a volatile access from becoming a flat one
__device__ __forceinline__ f32x2_w4 w4_ld2(const float* p) { return *(const volatile __attribute__((address_space(3))) f32x2_w4*)p; }
__device__ __forceinline__ void w4_st2(float* p, f32x2_w4 v) { *(volatile __attribute__((address_space(3))) f32x2_w4*)p = v; }
__device__ __forceinline__ int w4_opaque(int idx) { asm volatile("" : "+v"(idx)); return idx; }
constexpr unsigned W4_OOB = 0xfffffff0u;
__device__ __forceinline__ int w4_skew(int g) { return 2 * ((g >> 1) & 1); }
}  // namespace

// ABL (timing-only builds, results wrong): 1 no V transform, 4 no MFMAs, 8 no filter DMA, 32 no raw loads
template <int ABL, bool SPADE = false>
__global__ __launch_bounds__(512, 2) void wino4_kernel(const Wino4Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ub = smem;                           // [2][18][UPP]
    float* const Vb = smem + 2 * UBUF;                // [2][18][VPP]
    float* const Rb = Vb + 2 * VBUF;                  // [2][8][PL]
    float* const Bs = Rb + 2 * RAWBUF;                // [BIAS4]

    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, tg = wave >> 2;
    // MFMA role: A = U rows (couts 16 cg + l16), B = V columns (tiles 16 tg + l16), k = channel kq of the chunk
    const int a_off = kq * 128 + ((2 * (16 * cg + l16) + 32 * kq) & 127);
    const int b_off = kq * 64 + ((2 * (16 * tg + l16) + 32 * kq) & 63);
    // The two roles keep their per-thread state in the SAME registers (a wave is a T wave or an S wave for the whole launch; the role
    // branches are wave-uniform): ro[] = role constants, sc[] = T: patch rows d (30), B^T d (18), one output row (6) | S: the NIT staged float4
    // T role (waves 0-3): rows 3 rh .. 3 rh + 2 of V for tile m_t, channel kq_t of the chunk
    //   ro[0] = v_lo, ro[1] = v_hi (patch rows 0-3 / 4-5: the tile-row groups ty / ty + 1 carry different skews), ro[2] = t_dst
    // S role (waves 4-7): NIT (pixel, quad) items of the 18 x 34 x 8-channel raw block
    //   ro[it] = LDS offset of item it, ro[NIT + it] = byte offset of its pixel in the current block (W4_OOB: outside the image)
    const int rh = wave & 1;
    const int st = tid - 256;
    int ro[2 * NIT];
    float sc[54];
    if (wave < 4) {
        const int m_t = lane & 31, kq_t = 2 * ((wave >> 1) & 1) + (lane >> 5);
        const int ty_t = m_t >> 3, tx_t = m_t & 7;
        ro[0] = kq_t * PL + 4 * ty_t * RWP + w4_skew(ty_t) + 4 * tx_t;
        ro[1] = kq_t * PL + 4 * ty_t * RWP + w4_skew(ty_t + 1) + 4 * tx_t;
        ro[2] = kq_t * 64 + ((2 * m_t + 32 * kq_t) & 63);
#pragma unroll
        for (int it = 3; it < 2 * NIT; ++it) ro[it] = 0;
    } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = st + 256 * it, pi = idx >> 1, ry = pi / RBW, rx = pi - ry * RBW;
            // an item beyond the block (the tail of the last round) stores into the unused columns 36-39 of row 0: no branch around a store
            ro[it] = idx < NITEM ? 4 * (idx & 1) * PL + ry * RWP + w4_skew(ry >> 2) + rx : 36 + (lane & 3);
            ro[NIT + it] = (int)W4_OOB;
        }
    }
#pragma unroll
    for (int k = 0; k < 54; ++k) sc[k] = 0.f;
#define W4_VLO ro[0]
#define W4_VHI ro[1]
#define W4_TDST ro[2]
#define W4_SL(it) ro[it]
#define W4_XG(it) ro[NIT + (it)]
#define W4_D(ii, j) sc[6 * (ii) + (j)]
#define W4_R(a, j) sc[30 + 6 * (a) + (j)]
#define W4_VO(k) sc[48 + (k)]
#define W4_XR(it, c) sc[4 * (it) + (c)]

    const int grid = gridDim.x;
    const int rb0 = mrdis_xcd_remap(blockIdx.x, grid);
    const int nmine = (p.nblk - rb0 + grid - 1) / grid;           // host: grid <= nblk
    const int nch = p.Cin / KC4;                                  // host: Cin % 8 == 0
    const int total = nmine * nch;
    auto decode = [&](int j, int& n, int& oy0, int& ox0, int& cot) {
        int b = rb0 + j * grid;
        cot = b % p.coTiles; b /= p.coTiles;
        const int bx = b % p.nbx; b /= p.nbx;
        const int by = b % p.nby;
        n = b / p.nby; oy0 = 16 * by; ox0 = 32 * bx;
    };

    // ---- S: raw-block cursor.  Every global load is a buffer load whose offset is W4_OOB where there is nothing to read
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    int rj = 0, rc = 0;
    auto raw_block = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) W4_XG(it) = (int)W4_OOB;
        if (rj < nmine) {
            int n, oy0, ox0, cot; decode(rj, n, oy0, ox0, cot);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = st + 256 * it, pi = idx >> 1, ry = pi / RBW, rx = pi - ry * RBW;
                const int h = oy0 - 1 + ry, w_ = ox0 - 1 + rx;
                if (idx < NITEM && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                    W4_XG(it) = (int)(4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldin + 4 * (idx & 1)));      // host: < 2^30 elements
            }
        }
    };
    auto load_raw1 = [&](int it, unsigned c0b) {      // c0b: byte offset of the double chunk's first channel
        if (ABL & 32) { W4_XR(it, 0) = 0.f; W4_XR(it, 1) = 0.f; W4_XR(it, 2) = 0.f; W4_XR(it, 3) = 0.f; return; }
        const unsigned xg = (unsigned)W4_XG(it);
        const u32x4_w4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(xg != W4_OOB ? xg + c0b : W4_OOB), 0, 0);
        W4_XR(it, 0) = __uint_as_float(v.x); W4_XR(it, 1) = __uint_as_float(v.y); W4_XR(it, 2) = __uint_as_float(v.z); W4_XR(it, 3) = __uint_as_float(v.w);
    };
    auto raw_advance = [&]() { if (++rc == nch / 2) { rc = 0; ++rj; raw_block(); } };
    auto raw_store1 = [&](float* Rw, int it) {
        float* d_ = Rw + W4_SL(it); d_[0] = W4_XR(it, 0); d_[PL] = W4_XR(it, 1); d_[2 * PL] = W4_XR(it, 2); d_[3 * PL] = W4_XR(it, 3);
    };

    // ---- S: filter cursor (the image piece of (cout tile, chunk) is one contiguous 36 KB block).  The copies are `global_load_lds_dwordx4`
    // written as inline assembly: through the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of EVERY later LDS read of the kernel (it
    // cannot tell which LDS bytes the copy lands on), which serialises the pipeline; inline assembly is invisible to its wait-count pass (the
    // waits it inserts for the raw-block loads only become stricter, never wrong: vmcnt retires in order), and the S waves wait for their
    // own copies explicitly before the iteration's barrier.  There is no destination register a late write-back could corrupt.
    int fj = 0, fc = 0;
    long long f_base = 0;                             // float offset of chunk 0 of the current block's cout tile (past the end: tile 0 again --
    auto filt_block = [&]() {                         //  a copy nobody reads -- rather than a branch around the copy)
        f_base = 0;
        if (fj < nmine) { int n, oy0, ox0, cot; decode(fj, n, oy0, ox0, cot); f_base = (long long)cot * nch * UBUF; }
    };
    const float* f_chunk = p.u_img;                   // wave-uniform: image piece of the NEXT chunk
    const unsigned f_voff = 16u * (unsigned)lane + 1024u * (unsigned)(wave & 3);        // this lane's bytes inside piece q = (wave - 4) + 4 k
    const unsigned lds_u = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Ub;
    auto filt_next = [&]() {
        f_chunk = p.u_img + f_base + (long long)fc * UBUF;
        if (++fc == nch) { fc = 0; ++fj; filt_block(); }
    };
    auto dma1 = [&](int buf, int k) {                 // piece q = (wave - 4) + 4 k of 36: 1 KiB, lane l lands at U[buf] + 1024 q + 16 l bytes
        if (ABL & 8) return;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_u + 4u * (unsigned)(buf * UBUF) + 1024u * (unsigned)((wave & 3) + 4 * k));
        const float* src = f_chunk + 1024 * k;        // + 4 KiB per round of the four S waves
        unsigned keep;                                 // M0 is the compiler's: written and restored inside the one statement that reads it
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(m0v), "v"(f_voff), "s"(src) : "memory");
    };

    // ---- T: input transform pieces
    auto v_row = [&](const float* Rr, int ii, auto RH_) {           // patch row i = RH + ii (Rr: the chunk's channel planes of the raw block)
        const int row = decltype(RH_)::value + ii;                  // compile-time after unrolling
        const float* src = Rr + (row < 4 ? W4_VLO : W4_VHI) + row * RWP;
        const f32x2_w4 a = w4_ld2(src), b = w4_ld2(src + 2), c = w4_ld2(src + 4);
        W4_D(ii, 0) = a.x; W4_D(ii, 1) = a.y; W4_D(ii, 2) = b.x; W4_D(ii, 3) = b.y; W4_D(ii, 4) = c.x; W4_D(ii, 5) = c.y;
    };
    auto v_col = [&](int j, auto RH_) {               // three rows of B^T d in column j: B^T rows 0-2 on patch rows 0-4, rows 3-5 on patch rows 1-5
        const float e0 = W4_D(0, j), e1 = W4_D(1, j), e2 = W4_D(2, j), e3 = W4_D(3, j), e4 = W4_D(4, j);
        if constexpr (decltype(RH_)::value == 0) {
            const float a = fmaf(-4.f, e2, e4), b = fmaf(-4.f, e1, e3);
            W4_R(0, j) = fmaf(4.f, e0, fmaf(-5.f, e2, e4)); W4_R(1, j) = a + b; W4_R(2, j) = a - b;
        } else {
            const float c = e3 - e1, e = e2 - e0;
            W4_R(0, j) = fmaf(2.f, e, c); W4_R(1, j) = fmaf(-2.f, e, c); W4_R(2, j) = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
        }
    };
    auto v_rowop = [&](int a) {                       // (B^T d) B: the same combination along the row
        const float r0 = W4_R(a, 0), r1 = W4_R(a, 1), r2 = W4_R(a, 2), r3 = W4_R(a, 3), r4 = W4_R(a, 4), r5 = W4_R(a, 5);
        const float aa = fmaf(-4.f, r2, r4), bb = fmaf(-4.f, r1, r3), cc = r4 - r2, ee = r3 - r1;
        W4_VO(0) = fmaf(4.f, r0, fmaf(-5.f, r2, r4)); W4_VO(1) = aa + bb; W4_VO(2) = aa - bb;
        W4_VO(3) = fmaf(2.f, ee, cc); W4_VO(4) = fmaf(-2.f, ee, cc); W4_VO(5) = fmaf(4.f, r1, fmaf(-5.f, r3, r5));
    };
    auto v_put = [&](float* Vn, int a, auto RH_) {    // row 3 RH + a of V: point pairs 3 (3 RH + a) + 0..2
        constexpr int RH = decltype(RH_)::value;
        float* vp = Vn + (3 * (3 * RH + a)) * VPP;
        w4_st2(vp, f32x2_w4{W4_VO(0), W4_VO(1)});
        w4_st2(vp + VPP, f32x2_w4{W4_VO(2), W4_VO(3)});
        w4_st2(vp + 2 * VPP, f32x2_w4{W4_VO(4), W4_VO(5)});
    };

    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ABL & 64: in-kernel stamps (the stamps go to a buffer nothing else reads; the build is for diagnosis only, its timings carry ~10 % of overhead)
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

    // ---- prologue: U(0), V(0) and the raw double chunks 0 and 1 in LDS
    if ((p.prio == 1 && wave >= 4) || (p.prio == 2 && wave < 4)) __builtin_amdgcn_s_setprio(1);
    for (int c = tid; c < BIAS4; c += NT4) Bs[c] = (p.bias != nullptr && c < p.Cout) ? p.bias[c] : 0.f;
    if (wave >= 4) {
        raw_block(); filt_block();
#pragma unroll
        for (int it = 0; it < NIT; ++it) load_raw1(it, 32u * (unsigned)rc);
        raw_advance();
        filt_next();
#pragma unroll
        for (int k = 0; k < 9; ++k) dma1(0, k);
#pragma unroll
        for (int it = 0; it < NIT; ++it) raw_store1(Rb, it);
#pragma unroll
        for (int it = 0; it < NIT; ++it) load_raw1(it, 32u * (unsigned)rc);      // double chunk 1
        raw_advance();
#pragma unroll
        for (int it = 0; it < NIT; ++it) raw_store1(Rb + RAWBUF, it);
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the filter pieces have landed
    }
    __syncthreads();
    if (wave < 4) {
        const float* Rr = Rb;
        float* Vn = Vb + W4_TDST;
        if (rh == 0) {
#pragma unroll
            for (int ii = 0; ii < 5; ++ii) v_row(Rr, ii, IC4<0>{});
#pragma unroll
            for (int j = 0; j < 6; ++j) v_col(j, IC4<0>{});
#pragma unroll
            for (int a = 0; a < 3; ++a) { v_rowop(a); v_put(Vn, a, IC4<0>{}); }
        } else {
#pragma unroll
            for (int ii = 0; ii < 5; ++ii) v_row(Rr, ii, IC4<1>{});
#pragma unroll
            for (int j = 0; j < 6; ++j) v_col(j, IC4<1>{});
#pragma unroll
            for (int a = 0; a < 3; ++a) { v_rowop(a); v_put(Vn, a, IC4<1>{}); }
        }
    }
    __syncthreads();

    // one iteration g of parity P = g & 1 in role ROLE (0 / 1: T wave, rows 0-2 / 3-5 of V; 2: S wave)
    auto iteration = [&](auto P_, auto ROLE_, int g) {
        constexpr int P = decltype(P_)::value, ROLE = decltype(ROLE_)::value;
        const float* Ua = smem + w4_opaque(P * UBUF + a_off);
        const float* Va = smem + w4_opaque(2 * UBUF + P * VBUF + b_off);
        // T: V(g + 1) from channels 4 (P ^ 1) .. + 3 of raw double chunk (g + 1) / 2
        const float* Rr = smem + w4_opaque(2 * UBUF + 2 * VBUF + (((g + 1) >> 1) & 1) * RAWBUF + (P ^ 1) * 4 * PL);
        float* Vn = smem + w4_opaque(2 * UBUF + (P ^ 1) * VBUF + W4_TDST);
        // S: U(g + 1) by DMA; even g: loads of raw double chunk g / 2 + 2 (registers); odd g: their LDS stores, into the buffer whose block
        // (double chunk (g - 1) / 2) the T waves finished with in iteration g - 1.  A block ends after an odd iteration: nothing staged is live
        // in registers across the epilogue
        float* Rw = Rb + ((((g + 1) >> 1) + 1) & 1) * RAWBUF;
        unsigned c0b = 0;
        if constexpr (ROLE == 2) {
            filt_next();
            if (P == 0) c0b = 32u * (unsigned)rc;
        }

        stamp(1);
        f32x2_w4 av[3], bv[3];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            av[s_] = w4_ld2(Ua + s_ * UPP);
            bv[s_] = w4_ld2(Va + s_ * VPP);
        }
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            if (s_ + 2 < 18) {
                av[(s_ + 2) % 3] = w4_ld2(Ua + (s_ + 2) * UPP);
                bv[(s_ + 2) % 3] = w4_ld2(Va + (s_ + 2) * VPP);
            }
            if constexpr (ROLE < 2) {
                if (!(ABL & 1)) {            // (diagnosis: 128 no patch reads, 256 no V writes, 512 no transform arithmetic)
                    if (s_ < 5) { if (!(ABL & 128)) v_row(Rr, s_, IC4<ROLE>{}); }
                    else if (s_ < 11) { if (!(ABL & 512)) v_col(s_ - 5, IC4<ROLE>{}); }
                    else if (s_ < 17) {
                        if (((s_ - 11) & 1) == 0) { if (!(ABL & 512)) v_rowop((s_ - 11) >> 1); }
                        else if (!(ABL & 256)) v_put(Vn, (s_ - 11) >> 1, IC4<ROLE>{});
                    }
                }
            } else {
                if (s_ < 9) dma1(P ^ 1, s_);
                else if (P == 0) { if (s_ - 9 < NIT) load_raw1(s_ - 9, c0b); }
                else if (s_ >= 12 && s_ - 12 < NIT) raw_store1(Rw, s_ - 12);
            }
            const int c_ = s_ % 3;
            if (!(ABL & 4)) {
                acc[2 * s_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].x, bv[c_].x, acc[2 * s_], 0, 0, 0);
                acc[2 * s_ + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].y, bv[c_].y, acc[2 * s_ + 1], 0, 0, 0);
            } else { acc[2 * s_][0] += av[c_].x * bv[c_].x; acc[2 * s_ + 1][0] += av[c_].y * bv[c_].y; }
#pragma unroll
            for (int g_ = 0; g_ < 2; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // then at most one vector-memory read,
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // two LDS reads,
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);      // two LDS writes
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // and six VALU instructions in its shadow
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(2);
        if constexpr (ROLE == 2) {
            if (P == 0) { raw_advance(); __builtin_amdgcn_s_waitcnt(0x0F70 | NIT); }       // the DMA pieces are older than the NIT raw loads
            else __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        stamp(3);
        __syncthreads();
        stamp(4);
    };

    // the whole chunk loop once per role (a role branch INSIDE the loop makes the 144 accumulators a three-way phi at the loop header: hipcc
    // then keeps two copies of them and spills ~350 registers)
    auto run = [&](auto ROLE_) {
    int mj = 0, mc = 0;
    for (int g = 0; g < total; g += 2) {
        if constexpr (SPADE) {
            if (mc + 2 == nch && p.z_bytes != 0) {
                // The block's last two chunks: touch the z values its epilogue will read, so that they come into L2 / the Infinity Cache while the MFMAs of
                // these chunks run.  The epilogue's own loads cannot be hoisted over its 144 live accumulators (hoisting four of them spilled 25-63 registers)
                // and sit behind a bounds branch: eight exposed HBM round trips per block.  The prefetch needs a destination that is not a register (a load in
                // flight into a register the allocator has meanwhile given to another value would clobber it): one dword per lane by LDS-DMA into a
                // 256-byte strip per wave that nothing reads.  These loads are OLDER than every load of the two iterations, so the counted waits there cover them.
                int n_, oy0_, ox0_, cot_; decode(mj, n_, oy0_, ox0_, cot_);
                const int tile_ = 16 * tg + l16, oy_ = oy0_ + 4 * (tile_ >> 3), ox_ = ox0_ + 4 * (tile_ & 7);
                const int ch_ = 32 * cot_ + 8 * cg + 4 * (kq & 1);
                const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc((void*)p.z, 0, p.z_bytes, 0x00020000);
                const unsigned m0s = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)(Bs + BIAS4) + 256u * (unsigned)wave);
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const int row_ = oy_ + ip + 2 * (lane >> 5);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool ok_ = ch_ < p.C && row_ < p.H && ox_ + k < p.W;
                        const unsigned off_ = ok_ ? 4u * (unsigned)(((n_ * p.H + row_) * p.W + ox_ + k) * p.ldz + ch_) : W4_OOB;
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep) : "v"(off_), "s"(m0s), "s"(rs_z) : "memory");
                    }
                }
            }
        }
        iteration(IC4<0>{}, ROLE_, g); iteration(IC4<1>{}, ROLE_, g + 1);
        mc += 2;
        if (mc != nch) continue;
        mc = 0;
        stamp(5);
        // ---- epilogue of block mj: lane = tile 16 tg + l16, couts co0 + 16 cg + 4 kq + r;  Y = A^T M A,
        //      A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
        int n, oy0, ox0, cot; decode(mj, n, oy0, ox0, cot);
        ++mj;
        const int co0 = 64 * cot, tile = 16 * tg + l16;
        const int oy = oy0 + 4 * (tile >> 3), ox = ox0 + 4 * (tile & 7);
        const int co = co0 + 16 * cg + 4 * kq;
        if constexpr (SPADE) {
            // lane: channels ch .. ch + 3 of the block's 32; lower half-wave (kq 0, 1) = their gamma, upper half-wave (kq 2, 3) = their beta
            const int ch = 32 * cot + 8 * cg + 4 * (kq & 1);
            const bool ch_ok = ch < p.C;
            const int up = lane >> 5;                   // after the swap: lower lanes own output rows 0, 1 of the tile, upper lanes rows 2, 3
            const f32x4 mu = ch_ok ? *reinterpret_cast<const f32x4*>(p.mean + (long long)n * p.C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 rs = ch_ok ? *reinterpret_cast<const f32x4*>(p.rstd + (long long)n * p.C + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 bg = *reinterpret_cast<const f32x4*>(Bs + (ch_ok ? ch : 0));
            const f32x4 bb = *reinterpret_cast<const f32x4*>(Bs + (ch_ok ? p.C + ch : 0));
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                f32x4 yy[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = ip + 2 * h;
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
                    yy[h][0] = t[0] + s12 + s34; yy[h][1] = d12 + 2.f * d34; yy[h][2] = s12 + 4.f * s34; yy[h][3] = d12 + 8.f * d34 + t[5];
                }
                const int row = oy + ip + 2 * up;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 g, bt;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {       // (gamma row ip | beta row ip), (gamma row ip + 2 | beta row ip + 2) -> (gamma, gamma), (beta, beta)
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(yy[0][k][c]), __float_as_uint(yy[1][k][c]), false, false);
                        g[c] = __uint_as_float(sw[0]); bt[c] = __uint_as_float(sw[1]);
                    }
                    if (!(ch_ok && row < p.H && ox + k < p.W)) continue;
                    const long long pix = (long long)(n * p.H + row) * p.W + ox + k;
                    const f32x4 zv = *reinterpret_cast<const f32x4*>(p.z + pix * p.ldz + ch);
                    g = g + bg; bt = bt + bb;
                    const f32x4 mixv = (zv - mu) * rs * (g + 1.f) + bt;
                    // gamma is read again only by the backward pass: non-temporal; mix feeds the next convolution: non-temporal only when the tensor is beyond
                    // what stays cached anyway (nt_out; debug_mode 2006 / 2007: neither / both, for A/B)
                    if (p.nt_out & 1) __builtin_nontemporal_store(mixv, reinterpret_cast<f32x4*>(p.out + pix * p.ldout + ch));
                    else *reinterpret_cast<f32x4*>(p.out + pix * p.ldout + ch) = mixv;
                    if (p.nt_out & 2) __builtin_nontemporal_store(g, reinterpret_cast<f32x4*>(p.gamma_out + pix * p.ldg + ch));
                    else *reinterpret_cast<f32x4*>(p.gamma_out + pix * p.ldg + ch) = g;
                }
            }
#pragma unroll
            for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const bool full = oy0 + 16 <= p.H && ox0 + 32 <= p.W && co0 + 64 <= p.Cout;          // block-uniform: no per-store tests
        const f32x4 slope = p.lrelu ? f32x4{0.2f, 0.2f, 0.2f, 0.2f} : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bs + (co < BIAS4 ? co : 0));
        float* const o00 = p.out + ((long long)(n * p.H + oy) * p.W + ox) * p.ldout + co;
        const long long rowp = (long long)p.W * p.ldout;
        // one output row at a time (6 + 4 live vectors instead of 24: the 144 accumulators stay where they are until the last row is out)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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
            f32x4 y[4];
            y[0] = t[0] + s12 + s34; y[1] = d12 + 2.f * d34; y[2] = s12 + 4.f * s34; y[3] = d12 + 8.f * d34 + t[5];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v = y[k] + b4;
                v = __builtin_elementwise_max(v, v * slope);          // LeakyReLU(0.2), or the identity
                float* dst = o00 + i * rowp + k * p.ldout;
                if (full || (co < p.Cout && oy + i < p.H && ox + k < p.W)) {
                    if (p.nt_out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                    else *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        stamp(6);
    }
    };
    if (wave >= 4) run(IC4<2>{});
    else if (rh) run(IC4<1>{});
    else run(IC4<0>{});
}


// =========================================================================== narrow form: 64 tiles x 32 couts per workgroup
// The layers with <= 32 couts -- the data gradient of the full-resolution gamma | beta convolution (64 -> 32 at 256x256), sp5.out (64 -> 32), ana.up_1
// (128 -> 32) -- ran on the phase-by-phase F(2x2) kernel for 32 couts (46 % matrix-pipe utilisation at 4/9 of the direct multiplies: 14 ms of the step).
// Here: 8 x 8 tiles (32 x 32 outputs) x 32 couts, wave = 16 tiles x 16 couts x 36 points as above (4 tile groups x 2 cout groups), 4-channel chunks.
// Every wave carries every role (twice the input-transform work per MFMA of the 64-cout form: 64 tiles per 32 couts): V rows 3 rh .. 3 rh + 2 of channel
// wave / 2 for the tile of its lane, its share of the raw block (one 16-byte pixel piece per item, single 4-channel chunks) and of the 18 filter pieces.
// LDS: U 2 x 18 KB + V 2 x 36 KB + raw 2 x 21.25 KB + bias = 152.5 KB.  Image (format 5): [32-cout tile][chunk][18 point pairs][4 kq][64 slots].
namespace {
constexpr int N_UPP = 256, N_UBUF = 18 * N_UPP;           // 4608 floats
constexpr int N_VPP = 512, N_VBUF = 18 * N_VPP;           // 9216 floats
constexpr int N_RB = 34, N_PL = N_RB * RWP, N_RAWBUF = 4 * N_PL;      // 34 x 34 raw block, 1360-float planes, 5440 floats per buffer
constexpr int N_NITEM = N_RB * N_RB, N_NIT = (N_NITEM + 511) / 512;  // 1156 pixel pieces, 3 per thread
constexpr size_t W4N_LDS = sizeof(float) * (2 * N_UBUF + 2 * N_VBUF + 2 * N_RAWBUF + BIAS4);
}  // namespace

template <int ABL>
__global__ __launch_bounds__(512, 2) void wino4n_kernel(const Wino4Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Ub = smem;                           // [2][18][N_UPP]
    float* const Vb = smem + 2 * N_UBUF;              // [2][18][N_VPP]
    float* const Rb = Vb + 2 * N_VBUF;                // [2][4][N_PL]
    float* const Bs = Rb + 2 * N_RAWBUF;

    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 1, tg = wave >> 1;
    const int a_off = kq * 64 + ((2 * (16 * cg + l16) + 32 * kq) & 63);
    const int b_off = kq * 128 + ((2 * (16 * tg + l16) + 32 * kq) & 127);
    // transform role: rows 3 rh .. 3 rh + 2 of V, channel kq_t of the chunk, tile = lane (8 x 8 tiles)
    const int rh = wave & 1, kq_t = wave >> 1, ty_t = lane >> 3, tx_t = lane & 7;
    const int v_lo = kq_t * N_PL + 4 * ty_t * RWP + w4_skew(ty_t) + 4 * tx_t;
    const int v_hi = kq_t * N_PL + 4 * ty_t * RWP + w4_skew(ty_t + 1) + 4 * tx_t;
    const int t_dst = kq_t * 128 + ((2 * lane + 32 * kq_t) & 127);
    // staging role: N_NIT pixels of the 34 x 34 raw block (an item past the block stores into the unused columns 36-39 of row 0)
    int s_l[N_NIT]; unsigned xg[N_NIT];
#pragma unroll
    for (int it = 0; it < N_NIT; ++it) {
        const int idx = tid + 512 * it, ry = idx / N_RB, rx = idx - ry * N_RB;
        s_l[it] = idx < N_NITEM ? ry * RWP + w4_skew(ry >> 2) + rx : 36 + (lane & 3);
        xg[it] = W4_OOB;
    }

    const int grid = gridDim.x;
    const int rb0 = mrdis_xcd_remap(blockIdx.x, grid);
    const int nmine = (p.nblk - rb0 + grid - 1) / grid;
    const int nch = p.Cin / KC4;                                  // host: Cin % 8 == 0
    const int total = nmine * nch;
    auto decode = [&](int j, int& n, int& oy0, int& ox0, int& cot) {
        int b = rb0 + j * grid;
        cot = b % p.coTiles; b /= p.coTiles;
        const int bx = b % p.nbx; b /= p.nbx;
        const int by = b % p.nby;
        n = b / p.nby; oy0 = 32 * by; ox0 = 32 * bx;
    };
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    int rj = 0, rc = 0;
    auto raw_block = [&]() {
#pragma unroll
        for (int it = 0; it < N_NIT; ++it) xg[it] = W4_OOB;
        if (rj < nmine) {
            int n, oy0, ox0, cot; decode(rj, n, oy0, ox0, cot);
#pragma unroll
            for (int it = 0; it < N_NIT; ++it) {
                const int idx = tid + 512 * it, ry = idx / N_RB, rx = idx - ry * N_RB;
                const int h = oy0 - 1 + ry, w_ = ox0 - 1 + rx;
                if (idx < N_NITEM && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W)
                    xg[it] = 4u * (unsigned)(((n * p.H + h) * p.W + w_) * p.ldin);
            }
        }
    };
    float4 xr[N_NIT];
    auto load_raw1 = [&](int it, unsigned c0b) {
        if (ABL & 32) { xr[it] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
        const u32x4_w4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(xg[it] != W4_OOB ? xg[it] + c0b : W4_OOB), 0, 0);
        xr[it] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    auto raw_advance = [&]() { if (++rc == nch) { rc = 0; ++rj; raw_block(); } };
    auto raw_store1 = [&](float* Rw, int it) {
        float* d_ = Rw + s_l[it]; d_[0] = xr[it].x; d_[N_PL] = xr[it].y; d_[2 * N_PL] = xr[it].z; d_[3 * N_PL] = xr[it].w;
    };
    int fj = 0, fc = 0;
    long long f_base = 0;
    auto filt_block = [&]() {
        f_base = 0;
        if (fj < nmine) { int n, oy0, ox0, cot; decode(fj, n, oy0, ox0, cot); f_base = (long long)cot * nch * N_UBUF; }
    };
    const float* f_chunk = p.u_img;
    const unsigned f_voff = 16u * (unsigned)lane + 1024u * (unsigned)wave;      // piece q = wave + 8 k of 18
    const unsigned lds_u = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Ub;
    auto filt_next = [&]() {
        f_chunk = p.u_img + f_base + (long long)fc * N_UBUF;
        if (++fc == nch) { fc = 0; ++fj; filt_block(); }
    };
    auto dma1 = [&](int buf, int k) {
        if ((ABL & 8) || wave + 8 * k >= 18) return;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_u + 4u * (unsigned)(buf * N_UBUF) + 1024u * (unsigned)(wave + 8 * k));
        const float* src = f_chunk + 2048 * k;
        unsigned keep;                                 // M0 is the compiler's: written and restored inside the one statement that reads it
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(m0v), "v"(f_voff), "s"(src) : "memory");
    };

    float d[5][6], r[3][6], vo[6];
    auto v_row = [&](const float* Rr, int ii, auto RH_) {
        const int row = decltype(RH_)::value + ii;
        const float* src = Rr + (row < 4 ? v_lo : v_hi) + row * RWP;
        const f32x2_w4 a = w4_ld2(src), b = w4_ld2(src + 2), c = w4_ld2(src + 4);
        d[ii][0] = a.x; d[ii][1] = a.y; d[ii][2] = b.x; d[ii][3] = b.y; d[ii][4] = c.x; d[ii][5] = c.y;
    };
    auto v_col = [&](int j, auto RH_) {
        const float e0 = d[0][j], e1 = d[1][j], e2 = d[2][j], e3 = d[3][j], e4 = d[4][j];
        if constexpr (decltype(RH_)::value == 0) {
            const float a = fmaf(-4.f, e2, e4), b = fmaf(-4.f, e1, e3);
            r[0][j] = fmaf(4.f, e0, fmaf(-5.f, e2, e4)); r[1][j] = a + b; r[2][j] = a - b;
        } else {
            const float c = e3 - e1, e = e2 - e0;
            r[0][j] = fmaf(2.f, e, c); r[1][j] = fmaf(-2.f, e, c); r[2][j] = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
        }
    };
    auto v_rowop = [&](int a) {
        const float r0 = r[a][0], r1 = r[a][1], r2 = r[a][2], r3 = r[a][3], r4 = r[a][4], r5 = r[a][5];
        const float aa = fmaf(-4.f, r2, r4), bb = fmaf(-4.f, r1, r3), cc = r4 - r2, ee = r3 - r1;
        vo[0] = fmaf(4.f, r0, fmaf(-5.f, r2, r4)); vo[1] = aa + bb; vo[2] = aa - bb;
        vo[3] = fmaf(2.f, ee, cc); vo[4] = fmaf(-2.f, ee, cc); vo[5] = fmaf(4.f, r1, fmaf(-5.f, r3, r5));
    };
    auto v_put = [&](float* Vn, int a, auto RH_) {
        constexpr int RH = decltype(RH_)::value;
        float* vp = Vn + (3 * (3 * RH + a)) * N_VPP;
        w4_st2(vp, f32x2_w4{vo[0], vo[1]}); w4_st2(vp + N_VPP, f32x2_w4{vo[2], vo[3]}); w4_st2(vp + 2 * N_VPP, f32x2_w4{vo[4], vo[5]});
    };

    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: U(0), V(0), raw chunks 0 and 1 in LDS, raw chunk 2 in registers
    for (int c = tid; c < BIAS4; c += NT4) Bs[c] = (p.bias != nullptr && c < p.Cout) ? p.bias[c] : 0.f;
    raw_block(); filt_block();
#pragma unroll
    for (int it = 0; it < N_NIT; ++it) load_raw1(it, 16u * (unsigned)rc);
    raw_advance();
    filt_next();
#pragma unroll
    for (int k = 0; k < 3; ++k) dma1(0, k);
#pragma unroll
    for (int it = 0; it < N_NIT; ++it) raw_store1(Rb, it);
#pragma unroll
    for (int it = 0; it < N_NIT; ++it) load_raw1(it, 16u * (unsigned)rc);
    raw_advance();
#pragma unroll
    for (int it = 0; it < N_NIT; ++it) raw_store1(Rb + N_RAWBUF, it);
#pragma unroll
    for (int it = 0; it < N_NIT; ++it) load_raw1(it, 16u * (unsigned)rc);       // chunk 2: stored in iteration 0
    raw_advance();
    __builtin_amdgcn_s_waitcnt(0x0F70 | N_NIT);
    __syncthreads();
    {
        float* Vn = Vb + t_dst;
        if (rh == 0) {
#pragma unroll
            for (int ii = 0; ii < 5; ++ii) v_row(Rb, ii, IC4<0>{});
#pragma unroll
            for (int j = 0; j < 6; ++j) v_col(j, IC4<0>{});
#pragma unroll
            for (int a = 0; a < 3; ++a) { v_rowop(a); v_put(Vn, a, IC4<0>{}); }
        } else {
#pragma unroll
            for (int ii = 0; ii < 5; ++ii) v_row(Rb, ii, IC4<1>{});
#pragma unroll
            for (int j = 0; j < 6; ++j) v_col(j, IC4<1>{});
#pragma unroll
            for (int a = 0; a < 3; ++a) { v_rowop(a); v_put(Vn, a, IC4<1>{}); }
        }
    }
    __syncthreads();

    auto iteration = [&](auto P_, auto RH_) {
        constexpr int P = decltype(P_)::value;
        const float* Ua = smem + w4_opaque(P * N_UBUF + a_off);
        const float* Va = smem + w4_opaque(2 * N_UBUF + P * N_VBUF + b_off);
        const float* Rr = smem + w4_opaque(2 * N_UBUF + 2 * N_VBUF + (P ^ 1) * N_RAWBUF);        // raw chunk g + 1
        float* Vn = smem + w4_opaque(2 * N_UBUF + (P ^ 1) * N_VBUF + t_dst);
        float* Rw = Rb + P * N_RAWBUF;                 // raw chunk g + 2 (in registers since iteration g - 1) goes where chunk g was
        filt_next();
        const unsigned c0b = 16u * (unsigned)rc;       // raw chunk g + 3
        f32x2_w4 av[3], bv[3];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) { av[s_] = w4_ld2(Ua + s_ * N_UPP); bv[s_] = w4_ld2(Va + s_ * N_VPP); }
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            if (s_ + 2 < 18) { av[(s_ + 2) % 3] = w4_ld2(Ua + (s_ + 2) * N_UPP); bv[(s_ + 2) % 3] = w4_ld2(Va + (s_ + 2) * N_VPP); }
            if (!(ABL & 1)) {
                if (s_ < 5) v_row(Rr, s_, RH_);
                else if (s_ < 11) v_col(s_ - 5, RH_);
                else if (s_ < 17) { if (((s_ - 11) & 1) == 0) v_rowop((s_ - 11) >> 1); else v_put(Vn, (s_ - 11) >> 1, RH_); }
            }
            // staging: the stores first (their data was loaded an iteration ago), then the filter pieces, then the next raw chunk's loads
            if (s_ < N_NIT) raw_store1(Rw, s_);
            else if (s_ < N_NIT + 3) dma1(P ^ 1, s_ - N_NIT);
            else if (s_ >= 9 && s_ - 9 < N_NIT) load_raw1(s_ - 9, c0b);
            const int c_ = s_ % 3;
            if (!(ABL & 4)) {
                acc[2 * s_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].x, bv[c_].x, acc[2 * s_], 0, 0, 0);
                acc[2 * s_ + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].y, bv[c_].y, acc[2 * s_ + 1], 0, 0, 0);
            } else { acc[2 * s_][0] += av[c_].x * bv[c_].x; acc[2 * s_ + 1][0] += av[c_].y * bv[c_].y; }
#pragma unroll
            for (int g_ = 0; g_ < 2; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        raw_advance();
        __builtin_amdgcn_s_waitcnt(0x0F70 | N_NIT);   // the filter pieces are older than the N_NIT raw loads
        __syncthreads();
    };

    auto run = [&](auto RH_) {
    int mj = 0, mc = 0;
    for (int g = 0; g < total; g += 2) {
        iteration(IC4<0>{}, RH_); iteration(IC4<1>{}, RH_);
        mc += 2;
        if (mc != nch) continue;                       // host: Cin % 8 == 0, so a block ends after an odd iteration
        mc = 0;
        {
            int n, oy0, ox0, cot; decode(mj, n, oy0, ox0, cot);
            ++mj;
            const int co0 = 32 * cot, tile = 16 * tg + l16;
            const int oy = oy0 + 4 * (tile >> 3), ox = ox0 + 4 * (tile & 7);
            const int co = co0 + 16 * cg + 4 * kq;
            const bool full = oy0 + 32 <= p.H && ox0 + 32 <= p.W && co0 + 32 <= p.Cout;
            const f32x4 slope = p.lrelu ? f32x4{0.2f, 0.2f, 0.2f, 0.2f} : f32x4{1.f, 1.f, 1.f, 1.f};
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(Bs + (co < BIAS4 ? co : 0));
            float* const o00 = p.out + ((long long)(n * p.H + oy) * p.W + ox) * p.ldout + co;
            const long long rowp = (long long)p.W * p.ldout;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
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
                f32x4 y[4];
                y[0] = t[0] + s12 + s34; y[1] = d12 + 2.f * d34; y[2] = s12 + 4.f * s34; y[3] = d12 + 8.f * d34 + t[5];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x4 v = y[k] + b4;
                    v = __builtin_elementwise_max(v, v * slope);
                    float* dst = o00 + i * rowp + k * p.ldout;
                    if (full || (co < p.Cout && oy + i < p.H && ox + k < p.W)) {
                        if (p.nt_out) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
                        else *reinterpret_cast<f32x4*>(dst) = v;
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    };
    if (rh) run(IC4<1>{}); else run(IC4<0>{});
}

int mrdis_run_wino4n(const float* x, int ldx, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu,
                     hipStream_t s, const float* u_img) {
    if (!u_img || (((uintptr_t)u_img) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (Ci % 8 != 0 || Ci < 16 || Co < 4 || Co % 4 != 0 || Co > 32 || ldx % 4 != 0 || ldy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    Wino4Params p{};
    p.in_bytes = (unsigned)(4LL * ((long long)(N * H) * W - 1) * ldx + 4LL * Ci);
    p.in = x; p.bias = bias; p.out = y; p.u_img = u_img;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldy;
    p.lrelu = lrelu; p.prio = mrdis_opt(MRDIS_OPT_MODE) == 1001 ? 1 : (mrdis_opt(MRDIS_OPT_MODE) == 1002 ? 2 : 0);
    { const long long mb = mrdis_opt(MRDIS_OPT_NT_MB); p.nt_out = (long long)N * H * W * ldy * 4 >= mb * 1000000LL ? 1 : 0; }
    p.nby = mrdis_cdiv(H, 32); p.nbx = mrdis_cdiv(W, 32);
    p.coTiles = mrdis_cdiv(Co, 32);
    const long long nblk = (long long)N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (mrdis_opt(MRDIS_OPT_WINO4) < 2 && (nblk < 192 || H < 32 || W < 32)) return MRDIS_EUNSUPPORTED;
    // A chunk is 16 bytes of every pixel, so a cache line of the input is touched by 4-8 successive chunks, an iteration apart: that re-use is served by
    // L2 / the Infinity Cache while the input (about) fits there (64 -> 32 at 128x128, 134 MB: 124 -> 100 us; 128 -> 32, 268 MB: 228 -> 157 us) and by HBM when it does
    // not (64 -> 32 at 256x256, 537 MB: 474 -> 511 us)
    if (mrdis_opt(MRDIS_OPT_WINO4) < 2 && (long long)N * H * W * Ci * 4 > 300000000LL) return MRDIS_EUNSUPPORTED;
    p.nblk = (int)nblk;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)wino4n_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4N_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = nblk < n_cu ? (int)nblk : n_cu;
    mrdis_count(MRDIS_CNT_WINO4N);
    MRDIS_LAUNCH(wino4n_kernel<0>, dim3(grid), dim3(NT4), W4N_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

#ifdef WINO4_ABLATIONS
static unsigned long long* g_w4_dbg = nullptr; static int g_w4_dbg_cap = 0;
extern "C" void mrdis_debug_wino4_stamps(void* buf, int cap_per_wave) { g_w4_dbg = (unsigned long long*)buf; g_w4_dbg_cap = cap_per_wave; }
#endif
// the kernel's shape limits (the policy -- which layers SHOULD take it -- is mrdis_wino_u_fmt + the grid test below)
int mrdis_run_wino4(const float* x, int ldx, const float* bias, float* y, int ldy, int N, int H, int W, int Ci, int Co, int lrelu,
                    hipStream_t s, const float* u_img) {
    if (!u_img || (((uintptr_t)u_img) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (Ci % 8 != 0 || Ci < 16 || Co < 4 || Co % 4 != 0 || Co > BIAS4 || ldx % 4 != 0 || ldy % 4 != 0 || ((((uintptr_t)x) | ((uintptr_t)y)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    Wino4Params p{};
    p.in_bytes = (unsigned)(4LL * ((long long)(N * H) * W - 1) * ldx + 4LL * Ci);
    p.in = x; p.bias = bias; p.out = y; p.u_img = u_img;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = Co; p.ldout = ldy;
    p.lrelu = lrelu; p.prio = mrdis_opt(MRDIS_OPT_MODE) == 1001 ? 1 : (mrdis_opt(MRDIS_OPT_MODE) == 1002 ? 2 : 0);
    { const long long mb = mrdis_opt(MRDIS_OPT_NT_MB); p.nt_out = (long long)N * H * W * ldy * 4 >= mb * 1000000LL ? 1 : 0; }
    p.nby = mrdis_cdiv(H, 16); p.nbx = mrdis_cdiv(W, 32);
    p.coTiles = mrdis_cdiv(Co, 64);
    const long long nblk = (long long)N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    // a persistent workgroup per CU: below ~3/4 of the chip (or on maps that leave most of a 16 x 32 block empty) the F(2x2) kernel's
    // 16 x 16 blocks fill it better
    if (mrdis_opt(MRDIS_OPT_WINO4) < 2 && (nblk < 192 || H < 16 || W < 32)) return MRDIS_EUNSUPPORTED;
    p.nblk = (int)nblk;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)wino4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
#ifdef WINO4_ABLATIONS
#define W4A(a) hipFuncSetAttribute((const void*)wino4_kernel<a>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_LDS);
        W4A(1) W4A(4) W4A(5) W4A(8) W4A(32) W4A(41) W4A(45) W4A(64) W4A(65) W4A(72) W4A(96) W4A(105) W4A(192) W4A(320) W4A(576) W4A(448) W4A(832)
#undef W4A
#endif
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = nblk < n_cu ? (int)nblk : n_cu;
#ifdef WINO4_ABLATIONS
    const int abl = (int)mrdis_opt(MRDIS_OPT_MODE);          // debug_mode doubles as the ablation selector in this build
    p.dbg = g_w4_dbg; p.dbg_cap = g_w4_dbg_cap;
#define W4A(a) if (abl == a) { MRDIS_LAUNCH(wino4_kernel<a>, dim3(grid), dim3(NT4), W4_LDS, s, p); MRDIS_CHECK_LAUNCH(); return MRDIS_OK; }
    W4A(1) W4A(4) W4A(5) W4A(8) W4A(32) W4A(41) W4A(45) W4A(64) W4A(65) W4A(72) W4A(96) W4A(105) W4A(192) W4A(320) W4A(576) W4A(448) W4A(832)
#undef W4A
#endif
    mrdis_count(MRDIS_CNT_WINO4);
    MRDIS_LAUNCH(wino4_kernel<0>, dim3(grid), dim3(NT4), W4_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// SPADE-fused form (see Wino4Params): x = si_out (N, H, W, Ci), image of the fused gamma | beta filter [9][Ci][2 C] in the SPADE cout order, bias (2 C);
// z (N, H, W, C) with its instance statistics; writes mix and gamma.  MRDIS_EUNSUPPORTED: the caller takes the F(2x2) form / the two-step path.
int mrdis_run_wino4_spade(const float* x, int ldx, const float* bias, const float* z, int ldz, const float* mean, const float* rstd,
                          float* mix, int ldmix, float* gamma, int ldg, int N, int H, int W, int Ci, int C, hipStream_t s, const float* u_img) {
    if (!u_img || (((uintptr_t)u_img) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if (C < 8 || C % 8 != 0 || 2 * C > BIAS4 || Ci % 8 != 0 || Ci < 16 || ldx % 4 != 0 || ldz % 4 != 0 || ldmix % 4 != 0 || ldg % 4 != 0) return MRDIS_EUNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)z) | ((uintptr_t)mix) | ((uintptr_t)gamma) | ((uintptr_t)mean) | ((uintptr_t)rstd)) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)N * H * W * ldx >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    Wino4Params p{};
    p.in_bytes = (unsigned)(4LL * ((long long)(N * H) * W - 1) * ldx + 4LL * Ci);
    p.in = x; p.bias = bias; p.out = mix; p.u_img = u_img;
    p.N = N; p.H = H; p.W = W; p.Cin = Ci; p.ldin = ldx; p.Cout = 2 * C; p.ldout = ldmix;
    p.prio = mrdis_opt(MRDIS_OPT_MODE) == 1001 ? 1 : (mrdis_opt(MRDIS_OPT_MODE) == 1002 ? 2 : 0);
    p.z = z; p.ldz = ldz; p.mean = mean; p.rstd = rstd; p.gamma_out = gamma; p.ldg = ldg; p.C = C;
    { const long long md = mrdis_opt(MRDIS_OPT_MODE), mb = mrdis_opt(MRDIS_OPT_NT_MB);
      const int big = (long long)N * H * W * ldmix * 4 >= mb * 1000000LL ? 1 : 0;
      p.nt_out = md == 2006 ? 0 : (md == 2007 ? 3 : (md == 2008 ? 2 : (2 | big))); }
    { const long long zb = 4LL * ((long long)(N * H) * W - 1) * ldz + 4LL * C; p.z_bytes = (zb < 0xffffffe0LL && mrdis_opt(MRDIS_OPT_MODE) != 2005) ? (unsigned)zb : 0u; }      // (debug_mode 2005: no z prefetch, for A/B)
    p.nby = mrdis_cdiv(H, 16); p.nbx = mrdis_cdiv(W, 32);
    p.coTiles = mrdis_cdiv(C, 32);
    const long long nblk = (long long)N * p.nby * p.nbx * p.coTiles;
    if (nblk > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    if (mrdis_opt(MRDIS_OPT_WINO4) < 2 && (nblk < 192 || H < 16 || W < 32)) return MRDIS_EUNSUPPORTED;
    p.nblk = (int)nblk;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MRDIS_ELAUNCH;
        if (hipFuncSetAttribute((const void*)wino4_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int grid = nblk < n_cu ? (int)nblk : n_cu;
    mrdis_count(MRDIS_CNT_WINO4_SPADE);
    MRDIS_LAUNCH((wino4_kernel<0, true>), dim3(grid), dim3(NT4), W4_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// the format of a filter's Winograd image at policy level `level` (the value of option 'wino4': 0 | 1 | 2)
int mrdis_wino_u_fmt_at(int R, int S, int spadeC, int level) {
    if (!level) return 2;
    if (spadeC != 0) {                                 // fused gamma | beta filter (S = 2 C)
        // measured with the SPADE epilogue (tools/spade4_check.py, B = 32): 32 -> 2 x 32 at 256x256 532 -> 388 us, 64 -> 2 x 64 at 128x128 437 -> 301 us,
        // 128 -> 2 x 128 at 64x64 359 -> 232 us, at 32x32 (4 B images) 343 -> 224 us
        const int rmin_s = level >= 2 ? 16 : 32;
        return (R % 8 == 0 && R >= rmin_s && spadeC % 8 == 0 && spadeC >= 32 && 2 * spadeC <= BIAS4) ? 4 : 2;
    }
    // measured (tools/wino4_check.py, B = 32): 64 -> 128 at 128x128 349 -> 299 us forward / 323 -> 271 us data gradient, 128 -> 256 at 64x64 322 -> 210 /
    // 321 -> 205, 128 -> 64 at 64x64 85 -> 68 / 88 -> 74, 64 -> 64 at 128x128 171 -> 149 / 172 -> 158; 32 reduction channels (8 chunks per block: the
    // output transform + stores come round too often) 386 -> 387: no gain; < 64 couts leave half of the 64-cout tile empty
    const int rmin = level >= 2 ? 16 : 64;
    // <= 32 couts: the 32-cout forms (wino4r_kernel / wino4n_kernel) from 16 reduction channels on -- the data gradient of sp6.out (16 -> 32 at 256x256) 152 us on
    // the F(2x2) phase kernel, 118-126 us here
    if (R % 8 == 0 && R >= 16 && S <= 32 && S >= (level >= 2 ? 4 : 32) && S % 4 == 0) return 5;
    return (R % 8 == 0 && R >= rmin && S >= 64 && S % 4 == 0 && S <= BIAS4) ? 4 : 2;
}
int mrdis_wino_u_fmt(int R, int S, int spadeC) { return mrdis_wino_u_fmt_at(R, S, spadeC, (int)mrdis_opt(MRDIS_OPT_WINO4)); }
// can an image of a (R, S, spadeC) filter be of format `fmt` at all (under some value of the option)?  What the entry points check on the format a caller passes.
bool mrdis_wino_u_fmt_valid(int R, int S, int spadeC, int fmt) {
    return fmt == 2 || ((fmt == 4 || fmt == 5) && (fmt == mrdis_wino_u_fmt_at(R, S, spadeC, 1) || fmt == mrdis_wino_u_fmt_at(R, S, spadeC, 2)));
}
