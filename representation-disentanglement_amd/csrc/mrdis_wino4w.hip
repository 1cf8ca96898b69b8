// mrdis_wino4w.hip -- Winograd F(3x3, 4x4) weight gradient of the 3x3 / stride 1 / pad 1 layers for gfx950 (MI355X), fp32, NHWC
// (reference: autograd's convolution_backward (weight, bias) behind F.conv2d in CondConv2d._conv_forward, model.py:2104-2117).
//
// dW[ci][co] (3 x 3) = sum over 4 x 4 output tiles of the correlation of the 6 x 6 input patch with the 4 x 4 tile of dy
//                    = A^T [ sum_tiles (B^T d B) .* (G dY G^T) ] A,
// the F(3x3, 4x4) member of the family whose F(4x4, 3x3) member is the forward kernel (mrdis_wino4.hip): same interpolation points, the same
// B^T; 36 multiplies per tile and (ci, co) pair against 144 for the direct product and 64 for F(2x2) per the same 16 positions.  The reduction
// axis is the TILE index: 36 independent GEMMs dU_pt[ci][co] = sum_t V_pt[ci][t] Z_pt[t][co] with v_mfma_f32_16x16x4_f32 (k = four tiles).
// fp32 throughout; the transforms' growth (B^T: row sums <= 10) makes the result differ from the direct kernel's by ~1e-5 of the tensor's norm
// at the sizes of the step (direct: 1e-6; measured in tests/test_gpu_wino4.py; the goldens' per-tensor bar is 4e-4).
//
// Workgroup = 8 waves = a 32 x 64 block of (ci, co) and every `splits`-th group of 2 x 2 tiles (8 x 8 output positions); wave = 16 ci x 16 co x
// all 36 points = 144 accumulators.  One iteration = one tile group = ONE k-step (36 MFMAs per wave), one barrier; V and Z double-buffered:
//     iteration i:  36 MFMAs on V(i), Z(i)                                                                        all waves
//                   V(i+1) = B^T d B from the raw x block in LDS (72 VALU, 30 LDS reads, 9 writes per thread)     waves 0-3
//                   raw x block of group i+3: four 1-KiB LDS-DMA pieces per wave (no registers, no LDS store instructions)   waves 0-3
//                   Z(i+1) = G dY G^T from the 16 dy values loaded one iteration ago (90 VALU, 18 writes), loads of group i+2   waves 4-7
// G is applied WITHOUT its 1/4, 1/6, 1/24 factors (integers 1, 2, 4, 8 only); the factors multiply dU once, in the epilogue.  dbias = column sums
// of dy, taken from the Z waves' registers.  One slab [9][Ci][Co] per split, summed in a fixed order by wino_sum_slabs_kernel (bit-reproducible).
// LDS: V 2 x 18 KB + Z 2 x 36 KB + raw 3 x 16 KB = 156 KB.
#include "mrdis_wino.h"

namespace {
constexpr int W_NT = 512;
constexpr int W_VPP = 256, W_VBUF = 18 * W_VPP;        // [18 pairs][4 tiles][64 slots]: slot of (ci m, parity) = (2 m + parity + 32 tile) & 63
constexpr int W_ZPP = 512, W_ZBUF = 18 * W_ZPP;        // [18 pairs][4 tiles][128 slots]: slot of (co m, parity) = (2 m + parity + 32 tile) & 127
// raw x block: 10 x 10 pixels x 32 ci, copied by LDS-DMA: 16 pieces of 1 KiB = 128 pixel slots of 128 bytes (slots 100-127 are padding), three buffers (a copy has two
// iterations to land).  Slot p keeps global quad q (4 channels) at quad position q ^ (p & 4): the two tile columns of a group (4 pixels apart) land 16 banks apart
constexpr int W_PIX = 32, W_RAW = 128 * W_PIX;
constexpr int W_NXI = 4;                               // DMA pieces per V wave and iteration
constexpr size_t W4W_LDS = sizeof(float) * (2 * W_VBUF + 2 * W_ZBUF + 3 * W_RAW);
template <int V_> struct WIC { static constexpr int value = V_; };
typedef unsigned u32x4_ww __attribute__((ext_vector_type(4)));
typedef float f32x2_ww __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ww ww_ld2(const float* p) { return *(const volatile __attribute__((address_space(3))) f32x2_ww*)p; }
__device__ __forceinline__ void ww_st2(float* p, f32x2_ww v) { *(volatile __attribute__((address_space(3))) f32x2_ww*)p = v; }
__device__ __forceinline__ int ww_opaque(int idx) { asm volatile("" : "+v"(idx)); return idx; }
constexpr unsigned WW_OOB = 0xfffffff0u;
}  // namespace

struct Wino4WgradParams {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, H, W, Ci, ldx, Co, lddy;
    int ngy, ngx, ngroups;            // 8 x 8-position groups per image column / row, total
    int nCiB, nCoB, splits;
    int prio;                         // experiment: 1 = s_setprio 1 for waves 4-7, 2 = for waves 0-3
    int s_n, s_gy, s_gx;              // `splits` groups as (images, group rows, group columns): the cursors advance by it with carries
    unsigned long long* dbg; int dbg_cap;      // diagnostic build (-DWINO4_ABLATIONS): s_memtime stamps of workgroups 0-3
};

__global__ __launch_bounds__(512, 2) void wino4_wgrad_kernel(const Wino4WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const Vb = smem;                            // [2][W_VBUF]
    float* const Zb = smem + 2 * W_VBUF;               // [2][W_ZBUF]
    float* const Rb = Zb + 2 * W_ZBUF;                 // [2][W_RAW]

    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b_ = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);     // the (cib, cob) workgroups of a split read the same x / dy groups: same XCD, same L2
    const int cob = b_ % p.nCoB; b_ /= p.nCoB;
    const int cib = b_ % p.nCiB;
    const int split = b_ / p.nCiB;
    const int ci0 = 32 * cib, co0 = 64 * cob;
    // MFMA role: A = V rows (ci 16 wi + l16), B = Z columns (co 16 wo + l16), k = tile kq of the group
    const int wi = wave & 1, wo = wave >> 1;
    const int a_off = kq * 64 + ((2 * (16 * wi + l16) + 32 * kq) & 63);
    const int b_off = kq * 128 + ((2 * (16 * wo + l16) + 32 * kq) & 127);
    // transform roles: tile t = lane / 16 of the 2 x 2 group (ty = t / 2, tx = t % 2), channel lane % 16 of the wave's sixteen
    const int t_ = lane >> 4, ty = t_ >> 1, tx = t_ & 1;
    const int rh = wave & 1;                           // V waves (0-3): rows 3 rh .. 3 rh + 2 of V, ci 16 (wave / 2) + l16
    const int vc = 16 * ((wave >> 1) & 1) + l16;
    const int p0_ = (4 * ty) * 10 + 4 * tx;            // first pixel slot of the tile's patch; slot p0_ + d keeps channel vc at quad position (vc / 4) ^ (p0_ & 4) ^ (d & 4)
    const int v_src0 = p0_ * W_PIX + 4 * ((vc >> 2) ^ (p0_ & 4)) + (vc & 3);
    const int v_src1 = p0_ * W_PIX + 4 * ((vc >> 2) ^ (p0_ & 4) ^ 4) + (vc & 3);
    const int v_dst = t_ * 64 + ((2 * vc + 32 * t_) & 63);
    const int zc = 16 * (wave & 3) + l16;              // Z waves (4-7): co 16 (wave - 4) + l16
    const int z_dst = t_ * 128 + ((2 * zc + 32 * t_) & 127);

    struct Cur { int n, gy, gx; };
    auto advance = [&](Cur& c) {
        c.gx += p.s_gx; if (c.gx >= p.ngx) { c.gx -= p.ngx; ++c.gy; }
        c.gy += p.s_gy; if (c.gy >= p.ngy) { c.gy -= p.ngy; ++c.n; }
        c.n += p.s_n;
    };
    Cur xcur, dcur;
    { int t = split; xcur.gx = t % p.ngx; t /= p.ngx; xcur.gy = t % p.ngy; xcur.n = t / p.ngy; dcur = xcur; }
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)(4LL * ((long long)(p.N * p.H) * p.W - 1) * p.ldx + 4LL * p.Ci), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (unsigned)(4LL * ((long long)(p.N * p.H) * p.W - 1) * p.lddy + 4LL * p.Co), 0x00020000);

    // role state shared in registers: sc[] = V waves: two patch columns (10) + B^T d (18) + one V row (6);
    //                                        Z waves: two sets of 16 dy values (32) + G dY (24) + one Z row (6)
    float sc[62];
#pragma unroll
    for (int k = 0; k < 62; ++k) sc[k] = 0.f;
#define WW_D(ii, j) sc[5 * ((j) & 1) + (ii)]
#define WW_R(a, j) sc[10 + 6 * (a) + (j)]
#define WW_VO(k) sc[28 + (k)]
#define WW_DY(S, e) sc[16 * (S) + (e)]
#define WW_T(i, j) sc[32 + 4 * (i) + (j)]
#define WW_ZO(k) sc[56 + (k)]
    float bsum = 0.f;
    if ((p.prio == 1 && wave >= 4) || (p.prio == 2 && wave < 4)) __builtin_amdgcn_s_setprio(1);

    // ---- V waves: the raw x block by LDS-DMA (`buffer_load_dwordx4 ... offen lds`: 64 lanes x 16 bytes land at M0 + 16 lane; a lane whose pixel lies outside the
    //      image gets an offset beyond the descriptor's range, i.e. zeros: the convolution's zero padding).  Piece k = (wave & 3) + 4 j holds pixel slots 8 k .. 8 k + 7;
    //      lane l copies quad (l & 7) ^ (slot & 4) of slot 8 k + l / 8.  Inline assembly: hipcc would otherwise wait vmcnt(0) before every later LDS read.
    unsigned x_rel[W_NXI]; int x_yx[W_NXI];
#pragma unroll
    for (int j = 0; j < W_NXI; ++j) {
        const int slot = 8 * ((wave & 3) + 4 * j) + (lane >> 3), q = (lane & 7) ^ (slot & 4), ry = slot / 10, rx = slot - ry * 10;
        x_yx[j] = slot < 100 ? ((ry << 8) | rx) : -1;
        x_rel[j] = 4u * (unsigned)((ry * p.W + rx) * p.ldx + ci0 + 4 * q);
    }
    const unsigned lds_raw = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Rb;
    auto dma_x = [&](int buf, int j) {                 // piece j of this wave for the cursor's group into raw buffer `buf`; the cursor advances after the last piece
        const int h = 8 * xcur.gy - 1 + (x_yx[j] >> 8), w_ = 8 * xcur.gx - 1 + (x_yx[j] & 255);
        const bool ok = x_yx[j] >= 0 && xcur.n < p.N && (unsigned)h < (unsigned)p.H && (unsigned)w_ < (unsigned)p.W;
        const unsigned gb = 4u * (unsigned)(((xcur.n * p.H + 8 * xcur.gy - 1) * p.W + 8 * xcur.gx - 1) * p.ldx);     // wraps for halo origins; added mod 2^32
        const unsigned vo = ok ? gb + x_rel[j] : WW_OOB;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_raw + 4u * (unsigned)(buf * W_RAW) + 1024u * (unsigned)((wave & 3) + 4 * j));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(m0v), "s"(rs_x) : "memory");
        if (j == W_NXI - 1) advance(xcur);
    };
    // ---- V waves: B^T d B (the forward kernel's input transform: B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1])
    auto v_read = [&](const float* Rr, int j, auto RH_) {            // column j of patch rows RH .. RH + 4 (the patch is walked column by column: 10 live values, not 30)
#pragma unroll
        for (int ii = 0; ii < 5; ++ii) { const int d_ = (decltype(RH_)::value + ii) * 10 + j; WW_D(ii, j) = Rr[((d_ & 4) ? v_src1 : v_src0) + d_ * W_PIX]; }
    };
    auto v_col = [&](int j, auto RH_) {
        const float e0 = WW_D(0, j), e1 = WW_D(1, j), e2 = WW_D(2, j), e3 = WW_D(3, j), e4 = WW_D(4, j);
        if constexpr (decltype(RH_)::value == 0) {
            const float a = fmaf(-4.f, e2, e4), b = fmaf(-4.f, e1, e3);
            WW_R(0, j) = fmaf(4.f, e0, fmaf(-5.f, e2, e4)); WW_R(1, j) = a + b; WW_R(2, j) = a - b;
        } else {
            const float c = e3 - e1, e = e2 - e0;
            WW_R(0, j) = fmaf(2.f, e, c); WW_R(1, j) = fmaf(-2.f, e, c); WW_R(2, j) = fmaf(4.f, e0, fmaf(-5.f, e2, e4));
        }
    };
    auto v_rowop = [&](int a) {
        const float r0 = WW_R(a, 0), r1 = WW_R(a, 1), r2 = WW_R(a, 2), r3 = WW_R(a, 3), r4 = WW_R(a, 4), r5 = WW_R(a, 5);
        const float aa = fmaf(-4.f, r2, r4), bb = fmaf(-4.f, r1, r3), cc = r4 - r2, ee = r3 - r1;
        WW_VO(0) = fmaf(4.f, r0, fmaf(-5.f, r2, r4)); WW_VO(1) = aa + bb; WW_VO(2) = aa - bb;
        WW_VO(3) = fmaf(2.f, ee, cc); WW_VO(4) = fmaf(-2.f, ee, cc); WW_VO(5) = fmaf(4.f, r1, fmaf(-5.f, r3, r5));
    };
    auto v_put = [&](float* Vn, int a, auto RH_) {
        float* vp = Vn + (3 * (3 * decltype(RH_)::value + a)) * W_VPP;
        ww_st2(vp, f32x2_ww{WW_VO(0), WW_VO(1)}); ww_st2(vp + W_VPP, f32x2_ww{WW_VO(2), WW_VO(3)}); ww_st2(vp + 2 * W_VPP, f32x2_ww{WW_VO(4), WW_VO(5)});
    };
    // ---- Z waves: dy loads and the unscaled G dY G^T: rows of 24 G = [6 0 0 0; -4 -4 -4 -4; -4 4 -4 4; 1 2 4 8; 1 -2 4 -8; 0 0 0 24] with the row factors
    //      (6, -4, -4, 1, 1, 24) / 24 left out: w = (d0, e + o, e - o, p + q, p - q, d3), e = d0 + d2, o = d1 + d3, p = d0 + 4 d2, q = 2 d1 + 8 d3
    // dy(row e / 4, column e % 4) of this thread's tile in the cursor's group: the thread's offset inside a group is a launch constant, the group's origin and
    // the element's (row, column) offset are scalars (the latter rides in the instruction's scalar-offset operand); host: H, W multiples of 8, Co of 64
    const unsigned z_rel = 4u * (unsigned)(((4 * ty) * p.W + 4 * tx) * p.lddy + co0 + zc);
    const unsigned dy_row = 4u * (unsigned)(p.W * p.lddy), dy_col = 4u * (unsigned)p.lddy;
    unsigned z_vo = WW_OOB;
    auto dy_group = [&]() {                            // once per group, before its sixteen loads
        const unsigned gb = 4u * (unsigned)(((dcur.n * p.H + 8 * dcur.gy) * p.W + 8 * dcur.gx) * p.lddy);
        z_vo = dcur.n < p.N ? gb + z_rel : WW_OOB;
    };
    auto load_dy = [&](auto S_, int e) {               // advances the cursor after the sixteenth
        constexpr int S = decltype(S_)::value;
        if (e == 0) dy_group();
        WW_DY(S, e) = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dy, (int)z_vo, (int)((unsigned)(e >> 2) * dy_row + (unsigned)(e & 3) * dy_col), 0));
        if (e == 15) advance(dcur);
    };
    auto z_col = [&](auto S_, int j) {                 // column j of dY (4 values) -> column j of G dY (6 values)
        constexpr int S = decltype(S_)::value;
        const float d0 = WW_DY(S, j), d1 = WW_DY(S, 4 + j), d2 = WW_DY(S, 8 + j), d3 = WW_DY(S, 12 + j);
        bsum += (d0 + d1) + (d2 + d3);
        const float e = d0 + d2, o = d1 + d3, p_ = fmaf(4.f, d2, d0), q_ = 2.f * fmaf(4.f, d3, d1);
        WW_T(0, j) = d0; WW_T(1, j) = e + o; WW_T(2, j) = e - o; WW_T(3, j) = p_ + q_; WW_T(4, j) = p_ - q_; WW_T(5, j) = d3;
    };
    auto z_row = [&](int i) {                          // row i of (G dY) (4 values) -> row i of G dY G^T (6 values)
        const float d0 = WW_T(i, 0), d1 = WW_T(i, 1), d2 = WW_T(i, 2), d3 = WW_T(i, 3);
        const float e = d0 + d2, o = d1 + d3, p_ = fmaf(4.f, d2, d0), q_ = 2.f * fmaf(4.f, d3, d1);
        WW_ZO(0) = d0; WW_ZO(1) = e + o; WW_ZO(2) = e - o; WW_ZO(3) = p_ + q_; WW_ZO(4) = p_ - q_; WW_ZO(5) = d3;
    };
    auto z_put = [&](float* Zn, int i) {
        float* zp = Zn + (3 * i) * W_ZPP;
        ww_st2(zp, f32x2_ww{WW_ZO(0), WW_ZO(1)}); ww_st2(zp + W_ZPP, f32x2_ww{WW_ZO(2), WW_ZO(3)}); ww_st2(zp + 2 * W_ZPP, f32x2_ww{WW_ZO(4), WW_ZO(5)});
    };

    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] int n_stamp = 0;
    auto stamp = [&](int tag) {
#ifdef WINO4_ABLATIONS
        if (p.dbg != nullptr && blockIdx.x < 4 && n_stamp < p.dbg_cap) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.dbg[((long long)(blockIdx.x * 8 + wave)) * p.dbg_cap + n_stamp] = (t << 4) | (unsigned)tag;
            ++n_stamp;
        }
#endif
    };

    const int niter = (p.ngroups - split + p.splits - 1) / p.splits;
    // ---- prologue: V(0), Z(0) and raw(0..2) in LDS, dy(1) in set 1; cursors at raw(3), dy(2)
    if (wave < 4) {
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3)                 // raw(0), raw(1), raw(2) into the three buffers
#pragma unroll
            for (int j = 0; j < W_NXI; ++j) dma_x(g3, j);
        __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) load_dy(WIC<0>{}, e);
#pragma unroll
        for (int j = 0; j < 4; ++j) z_col(WIC<0>{}, j);
#pragma unroll
        for (int i = 0; i < 6; ++i) { z_row(i); z_put(Zb + z_dst, i); }
#pragma unroll
        for (int e = 0; e < 16; ++e) load_dy(WIC<1>{}, e);
    }
    __syncthreads();
    if (wave < 4) {
        if (rh == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) { v_read(Rb, j, WIC<0>{}); v_col(j, WIC<0>{}); }
#pragma unroll
            for (int a = 0; a < 3; ++a) { v_rowop(a); v_put(Vb + v_dst, a, WIC<0>{}); }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) { v_read(Rb, j, WIC<1>{}); v_col(j, WIC<1>{}); }
#pragma unroll
            for (int a = 0; a < 3; ++a) { v_rowop(a); v_put(Vb + v_dst, a, WIC<1>{}); }
        }
    }
    __syncthreads();

    int ring0 = 0, ring1 = 1;                          // i % 3, (i + 1) % 3
    // one iteration of parity P in role ROLE (0 / 1: V wave, rows 0-2 / 3-5 of V; 2: Z wave)
    auto iteration = [&](auto P_, auto ROLE_) {
        constexpr int P = decltype(P_)::value, ROLE = decltype(ROLE_)::value;
        const float* Va = smem + ww_opaque(P * W_VBUF + a_off);
        const float* Za = smem + ww_opaque(2 * W_VBUF + P * W_ZBUF + b_off);
        const float* Rr = smem + ww_opaque(2 * W_VBUF + 2 * W_ZBUF + ring1 * W_RAW);         // raw x block of group i + 1 (ring1 = (i + 1) % 3)
        float* Vn = smem + ww_opaque((P ^ 1) * W_VBUF + v_dst);
        float* Zn = smem + ww_opaque(2 * W_VBUF + (P ^ 1) * W_ZBUF + z_dst);
        stamp(1);
        f32x2_ww av[3], bv[3];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) { av[s_] = ww_ld2(Va + s_ * W_VPP); bv[s_] = ww_ld2(Za + s_ * W_ZPP); }
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            if (s_ + 2 < 18) { av[(s_ + 2) % 3] = ww_ld2(Va + (s_ + 2) * W_VPP); bv[(s_ + 2) % 3] = ww_ld2(Za + (s_ + 2) * W_ZPP); }
            if constexpr (ROLE < 2) {
                if (s_ < 6) v_read(Rr, s_, WIC<ROLE>{});
                if (s_ >= 1 && s_ < 7) v_col(s_ - 1, WIC<ROLE>{});
                else if (s_ >= 7 && s_ < 13) { if (((s_ - 7) & 1) == 0) v_rowop((s_ - 7) >> 1); else v_put(Vn, (s_ - 7) >> 1, WIC<ROLE>{}); }
                if (s_ < W_NXI) dma_x(ring0, s_);     // group i + 3 goes where group i was (ring0 = i % 3): read for the last time in iteration i - 1
            } else {
                // Z(i + 1) from dy set P ^ 1 (loaded one iteration ago); dy(i + 2) into set P, two values per step in the first eight steps
                if (s_ < 4) z_col(WIC<P ^ 1>{}, s_);
                else if (s_ < 16) { if (((s_ - 4) & 1) == 0) z_row((s_ - 4) >> 1); else z_put(Zn, (s_ - 4) >> 1); }
                if (s_ < 8) { load_dy(WIC<P>{}, 2 * s_); load_dy(WIC<P>{}, 2 * s_ + 1); }       // early: the first of them is needed at the top of the next iteration
            }
            const int c_ = s_ % 3;
            acc[2 * s_] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].x, bv[c_].x, acc[2 * s_], 0, 0, 0);
            acc[2 * s_ + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c_].y, bv[c_].y, acc[2 * s_ + 1], 0, 0, 0);
#pragma unroll
            for (int g_ = 0; g_ < 2; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(2);
        if constexpr (ROLE < 2) __builtin_amdgcn_s_waitcnt(0x0F70 | W_NXI);       // group i + 2's pieces (issued an iteration ago) have landed; this iteration's four may still fly
        ring0 = ring1; ring1 = ring1 == 2 ? 0 : ring1 + 1;
        stamp(3);
        __syncthreads();
        stamp(4);
    };
    auto run = [&](auto ROLE_) {
        // always pairs of iterations (a conditional second one makes the 144 accumulators a phi: two live copies, ~45 spilled registers); an odd count
        // runs one more group, which lies beyond the last one: its dy loads return zeros, so its Z is zero and it adds nothing
        for (int i = 0; i < niter; i += 2) { iteration(WIC<0>{}, ROLE_); iteration(WIC<1>{}, ROLE_); }
    };
    if (wave >= 4) run(WIC<2>{});
    else if (rh) run(WIC<1>{});
    else run(WIC<0>{});

    // ---- epilogue: dg = A^T (S dU S) A per lane, S = diag(1/4, -1/6, -1/6, 1/24, 1/24, 1) (the factors G was applied without), A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1];
    //      accumulator rows (4 kq + r) = ci, column l16 = co
    {
        const float sf[6] = {0.25f, -0.16666667f, -0.16666667f, 0.041666667f, 0.041666667f, 1.f};
        float* out = p.slab + (long long)split * 9 * p.Ci * p.Co;
        const int co = co0 + 16 * wo + l16, ci = ci0 + 16 * wi + 4 * kq;
        f32x4 tm[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const f32x4 m0 = acc[j] * sf[0], m1 = acc[6 + j] * sf[1], m2 = acc[12 + j] * sf[2], m3 = acc[18 + j] * sf[3], m4 = acc[24 + j] * sf[4], m5 = acc[30 + j] * sf[5];
            const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            tm[0][j] = (m0 + s12 + s34) * sf[j]; tm[1][j] = (d12 + 2.f * d34) * sf[j]; tm[2][j] = (s12 + 4.f * s34 + m5) * sf[j];
        }
        if (co < p.Co) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const f32x4 m0 = tm[a][0], m1 = tm[a][1], m2 = tm[a][2], m3 = tm[a][3], m4 = tm[a][4], m5 = tm[a][5];
                const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                const f32x4 g0 = m0 + s12 + s34, g1 = d12 + 2.f * d34, g2 = s12 + 4.f * s34 + m5;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    out[((long long)(3 * a + 0) * p.Ci + ci + r) * p.Co + co] = g0[r];
                    out[((long long)(3 * a + 1) * p.Ci + ci + r) * p.Co + co] = g1[r];
                    out[((long long)(3 * a + 2) * p.Ci + ci + r) * p.Co + co] = g2[r];
                }
            }
        }
    }
    if (p.bias_slab != nullptr && cib == 0 && wave >= 4) {
        // the four tile lanes (lane / 16) of a co meet by shuffles, fixed order
        float t = bsum;
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lane < 16 && co0 + zc < p.Co) p.bias_slab[(long long)split * p.Co + co0 + zc] = t;
    }
}

// Launches the F(3x3, 4x4) kernel on the plan of mrdis_run_wino_wgrad (x, dy, slab, bias_slab, N, H, W, Ci, Co, ldx, lddy filled in) and sets base.splits to
// the number of slabs written ([splits][9][Ci][Co], bias [splits][Co]); `max_splits` = what the workspace holds.  MRDIS_EUNSUPPORTED: the caller takes the F(2x2) kernels.
#ifdef WINO4_ABLATIONS
static unsigned long long* g_w4w_dbg = nullptr; static int g_w4w_dbg_cap = 0;
extern "C" void mrdis_debug_wino4w_stamps(void* buf, int cap_per_wave) { g_w4w_dbg = (unsigned long long*)buf; g_w4w_dbg_cap = cap_per_wave; }
#endif
int mrdis_launch_wino4_wgrad(WinoWgradParams& base, int max_splits, hipStream_t s) {
    const int opt = (int)mrdis_opt(MRDIS_OPT_WINO4);
    if (!opt || base.D != 0) return MRDIS_EUNSUPPORTED;
    if (base.Ci % 32 != 0 || base.Co % 64 != 0 || base.ldx % 4 != 0 || (((uintptr_t)base.x) & 15) != 0) return MRDIS_EUNSUPPORTED;
    if ((long long)base.N * base.H * base.W * base.ldx >= 0x3fffffffLL || (long long)base.N * base.H * base.W * base.lddy >= 0x3fffffffLL) return MRDIS_EUNSUPPORTED;
    Wino4WgradParams p{};
    p.prio = mrdis_opt(MRDIS_OPT_MODE) == 1001 ? 1 : (mrdis_opt(MRDIS_OPT_MODE) == 1002 ? 2 : 0);
    p.x = base.x; p.dy = base.dy; p.slab = base.slab; p.bias_slab = base.bias_slab;
    p.N = base.N; p.H = base.H; p.W = base.W; p.Ci = base.Ci; p.ldx = base.ldx; p.Co = base.Co; p.lddy = base.lddy;
    p.ngy = mrdis_cdiv(p.H, 8); p.ngx = mrdis_cdiv(p.W, 8);
    const long long ng = (long long)p.N * p.ngy * p.ngx;
    if (ng > 0x7fffffffLL) return MRDIS_EUNSUPPORTED;
    p.ngroups = (int)ng;
    p.nCiB = p.Ci / 32; p.nCoB = p.Co / 64;
    int splits = 256 / (p.nCiB * p.nCoB);
    if (splits < 1) splits = 1;
    if (splits > max_splits) splits = max_splits;
    if (splits > p.ngroups) splits = p.ngroups;
    // the kernel earns its prologue / epilogue only over enough iterations per workgroup, and only where the launch fills the chip
    if (p.H % 8 != 0 || p.W % 8 != 0) return MRDIS_EUNSUPPORTED;       // whole 8 x 8 groups only (the dy loads carry no per-element edge test)
    if (opt < 2 && (p.ngroups / splits < 24 || splits * p.nCiB * p.nCoB < 192 || p.H < 8 || p.W < 8)) return MRDIS_EUNSUPPORTED;
    p.splits = splits;
    { int t = splits; p.s_gx = t % p.ngx; t /= p.ngx; p.s_gy = t % p.ngy; p.s_n = t / p.ngy; }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wino4_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W4W_LDS) != hipSuccess) return MRDIS_EUNSUPPORTED;
        attr_set = true;
    }
#ifdef WINO4_ABLATIONS
    p.dbg = g_w4w_dbg; p.dbg_cap = g_w4w_dbg_cap;
#endif
    mrdis_count(MRDIS_CNT_WINO4_WGRAD);
    MRDIS_LAUNCH(wino4_wgrad_kernel, dim3(splits * p.nCiB * p.nCoB), dim3(W_NT), W4W_LDS, s, p);
    MRDIS_CHECK_LAUNCH();
    base.splits = splits;
    return MRDIS_OK;
}
