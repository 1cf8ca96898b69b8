// mrdis_elem.hip -- bandwidth-bound neighbours of the convolutions (gfx950, fp32):
// BatchNorm(train), InstanceNorm+SPADE modulation, bilinear resize, masked
// softmax, LeakyReLU backward, reconstruction error, max-pool, optimizer step.
// All tensors are NHWC "views" (rows = pixels, ld = row stride in floats).
// Reductions are two-level with a fixed summation order (partials in fp32 over
// short runs, combination in fp64) so every result is bit-reproducible.
#include "mrdis_common.h"
#include <mutex>
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

// ------------------------------------------------------------------ small vector helper
// Activations are stored as fp32 or bf16 (MRDIS_DT_BF16: BASELINE.json configs[2]); the arithmetic is fp32 either way:
// a kernel is templated on the storage type T and converts on load / store (bf16 stores round to nearest even).
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
template <int V> struct Vec;
template <> struct Vec<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
    __device__ __forceinline__ void store(float* p) const { p[0] = v[0]; }
    __device__ __forceinline__ void load(const __bf16* p) { v[0] = (float)p[0]; }
    __device__ __forceinline__ void store(__bf16* p) const { p[0] = (__bf16)v[0]; }
};
template <> struct Vec<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
    __device__ __forceinline__ void load(const __bf16* p) {
        const bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(p); v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    __device__ __forceinline__ void store(__bf16* p) const {
        bf16x4_t t; t[0] = (__bf16)v[0]; t[1] = (__bf16)v[1]; t[2] = (__bf16)v[2]; t[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4_t*>(p) = t;
    }
};
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const __bf16* p) {
    const bf16x4_t t = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }

template <typename T>
static inline bool vec4_ok(const T* p, int ld, int C) { return (C % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)p & (4 * sizeof(T) - 1)) == 0); }
// dtype of the activation views of an entry point: fp32 (MRDIS_DT_F32 and MRDIS_DT_F32_BF16M store fp32) or bf16
#define MRDIS_BY_DTYPE(dtype, CALL_F32, CALL_BF16) \
    ((dtype) == MRDIS_DT_BF16 ? (CALL_BF16) : (((dtype) == MRDIS_DT_F32 || (dtype) == MRDIS_DT_F32_BF16M) ? (CALL_F32) : MRDIS_EUNSUPPORTED))
static inline int ew_blocks(long long n) { long long b = (n + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }

#define EW_LOOP(total) for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < (total); idx += (long long)gridDim.x * blockDim.x)

// value (4 channels) at pixel (h, w) of the x2 bilinear resize (align_corners = False) of one image x (Hi x Wi, leading dimension ldx, pointer at the
// image's channel group), exactly as bilinear_up2_fwd_kernel forms and STORES it (same products, same association, rounded to T): the kernels that read a
// SPADE block's up-sampled input without its ever having been written (mrdis_instnorm_spade_bwd_up2 with xlo) take it from here.
template <typename T>
__device__ __forceinline__ void up2_value(const T* __restrict__ x, int ldx, int Hi, int Wi, int h, int w, float out[4]) {
    const int i = h >> 1, j = w >> 1;
    int r0, r1, c0, c1; float A0, A1, B0, B1;
    if (h & 1) { r0 = i; r1 = i < Hi - 1 ? i + 1 : Hi - 1; A0 = 0.75f; A1 = 0.25f; }
    else { r0 = i > 0 ? i - 1 : 0; r1 = i; A0 = i > 0 ? 0.25f : 0.f; A1 = i > 0 ? 0.75f : 1.f; }
    if (w & 1) { c0 = j; c1 = j < Wi - 1 ? j + 1 : Wi - 1; B0 = 0.75f; B1 = 0.25f; }
    else { c0 = j > 0 ? j - 1 : 0; c1 = j; B0 = j > 0 ? 0.25f : 0.f; B1 = j > 0 ? 0.75f : 1.f; }
    const float4 p00 = ld4(x + ((long long)r0 * Wi + c0) * ldx), p01 = ld4(x + ((long long)r0 * Wi + c1) * ldx);
    const float4 p10 = ld4(x + ((long long)r1 * Wi + c0) * ldx), p11 = ld4(x + ((long long)r1 * Wi + c1) * ldx);
    const float* a = &p00.x; const float* b = &p01.x; const float* c = &p10.x; const float* d = &p11.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (float)(T)(A0 * (B0 * a[k] + B1 * b[k]) + A1 * (B0 * c[k] + B1 * d[k]));
}

// ------------------------------------------------------------------ grouped column statistics
// rows of group g: [g*P, (g+1)*P).  part[((g*chunks + chunk)*2 + k)*C + c]
struct StatPlan { int chunks, rpb; };
static StatPlan stat_plan(int groups, long long P) {
    long long maxc = 1024 / (groups > 0 ? groups : 1); if (maxc < 1) maxc = 1;
    long long chunks = (P + 255) / 256; if (chunks > maxc) chunks = maxc; if (chunks < 1) chunks = 1;
    StatPlan s; s.chunks = (int)chunks; s.rpb = (int)((P + chunks - 1) / chunks);
    return s;
}
extern "C" size_t mrdis_norm_workspace(int groups, long long P, int C) {
    const StatPlan s = stat_plan(groups, P);
    return sizeof(float) * 2 * (size_t)groups * s.chunks * C + 64;
}

// MODE 0: (sum x, sum x^2)            a = x
// MODE 1: (sum dy, sum dy*xhat)       a = dy, b = x, stats (mean,rstd) per (group? no: per channel) -> BN bwd
// MODE 2: (sum dzh, sum dzh*zhat)     a = dout, b = z, c = gamma ; dzh = dout*(1+gamma)  -> SPADE/IN bwd
template <int MODE, typename T>
__global__ void stat_partial_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                    const T* __restrict__ g, int ldg, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, int stat_per_group,
                                    long long P, int C, int rpb, float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int grp = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const long long r0 = (long long)chunk * rpb;
    long long r1 = r0 + rpb; if (r1 > P) r1 = P;
    const long long gbase = (long long)grp * P;
    // lanes: C >= 64 -> 64 channels per pass; C < 64 and 64 % C == 0 -> 64/C rows per pass
    const int rpx = (C < 64 && (64 % C) == 0) ? 64 / C : 1;
    const int cw = (C < 64) ? C : 64;
    const int sub = threadIdx.x / cw, cl = threadIdx.x - sub * cw;
    const bool lane_ok = sub < rpx;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + cl;
        float s0 = 0.f, s1 = 0.f;
        if (lane_ok && c < C) {
            float mu = 0.f, rs = 1.f;
            if (MODE != 0) { const int si = stat_per_group ? grp * C + c : c; mu = mean[si]; rs = rstd[si]; }
            for (long long r = r0 + threadIdx.y * rpx + sub; r < r1; r += 4 * rpx) {
                const long long row = gbase + r;
                if (MODE == 0) { const float x = ld1(a + row * lda + c); s0 += x; s1 += x * x; }
                else if (MODE == 1) { const float d = ld1(a + row * lda + c); const float xh = (ld1(b + row * ldb + c) - mu) * rs; s0 += d; s1 += d * xh; }
                else { const float d = ld1(a + row * lda + c) * (1.f + ld1(g + row * ldg + c)); const float zh = (ld1(b + row * ldb + c) - mu) * rs; s0 += d; s1 += d * zh; }
            }
        }
        red[0][threadIdx.y][threadIdx.x] = s0; red[1][threadIdx.y][threadIdx.x] = s1;
        __syncthreads();
        if (threadIdx.y == 0 && sub == 0 && c < C) {
            float t0 = 0.f, t1 = 0.f;
            for (int y = 0; y < 4; ++y)
                for (int u = 0; u < rpx; ++u) { t0 += red[0][y][u * cw + cl]; t1 += red[1][y][u * cw + cl]; }
            float* dst = part + ((long long)(grp * chunks + chunk) * 2) * C;
            dst[c] = t0; dst[C + c] = t1;
        }
        __syncthreads();
    }
}

// Vector form of stat_partial_kernel (C % 4 == 0, 16-byte aligned views): a thread owns 4 consecutive channels
// and walks rows with float4 loads, `rows_pp` = 256 / (C/4) rows per pass of the block, 4 rows in flight per
// thread.  The scalar kernel keeps 4 bytes per lane in flight and reached 1.9 TB/s on the 268 MB maps; the
// statistics passes are pure streaming reads.  Block reduction in LDS, fixed order.
template <int MODE, typename T, bool INTERP = false>
__global__ __launch_bounds__(256) void stat_partial_vec_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                               const T* __restrict__ g, int ldg, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int stat_per_group,
                                                               long long P, int C, int rpb, float* __restrict__ part,
                                                               const T* __restrict__ xlo = nullptr, int ldxlo = 0, int Wlo = 0) {
    // xlo (MODE 2): b is not stored -- it is the x2 bilinear resize of the (group, P / 4 pixels, Wlo wide) map xlo and is interpolated here (up2_value)
    __shared__ float red[256][9];                      // [thread][8 sums], padded
    const int grp = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const long long r0 = (long long)chunk * rpb;
    long long r1 = r0 + rpb; if (r1 > P) r1 = P;
    const long long gbase = (long long)grp * P;
    const int cg_all = C >> 2;
    const int cg = cg_all < 256 ? cg_all : 256;        // channel groups handled per sweep
    const int rows_pp = 256 / cg;                      // rows per pass
    const int tid = threadIdx.x;
    const int rsub = tid / cg, cgi = tid - rsub * cg;
    const bool lane_ok = rsub < rows_pp;
    for (int cg0 = 0; cg0 < cg_all; cg0 += cg) {
        const int c = 4 * (cg0 + cgi);
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane_ok && c < C) {
            float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
            if (MODE != 0) {
                const int si = stat_per_group ? grp * C + c : c;
#pragma unroll
                for (int k = 0; k < 4; ++k) { mu[k] = mean[si + k]; rs[k] = rstd[si + k]; }
            }
            auto one = [&](long long r) {
                const long long row = gbase + r;
                const float4 av = ld4(a + row * lda + c);
                const float aa[4] = {av.x, av.y, av.z, av.w};
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s0[k] += aa[k]; s1[k] += aa[k] * aa[k]; }
                } else {
                    float bb[4];
                    if (MODE == 2 && INTERP) {
                        const int Wo_ = 2 * Wlo, h_ = (int)(r / Wo_), w_ = (int)(r - (long long)h_ * Wo_);
                        up2_value<T>(xlo + (long long)grp * (P >> 2) * ldxlo + c, ldxlo, (int)(P / (4LL * Wlo)), Wlo, h_, w_, bb);
                    } else {
                        const float4 bv = ld4(b + row * ldb + c);
                        bb[0] = bv.x; bb[1] = bv.y; bb[2] = bv.z; bb[3] = bv.w;
                    }
                    float gg[4] = {0.f, 0.f, 0.f, 0.f};
                    if (MODE == 2) { const float4 gv = ld4(g + row * ldg + c); gg[0] = gv.x; gg[1] = gv.y; gg[2] = gv.z; gg[3] = gv.w; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d = (MODE == 2) ? aa[k] * (1.f + gg[k]) : aa[k];
                        const float xh = (bb[k] - mu[k]) * rs[k];
                        s0[k] += d; s1[k] += d * xh;
                    }
                }
            };
            long long r = r0 + rsub;
            for (; r + 3LL * rows_pp < r1; r += 4LL * rows_pp) { one(r); one(r + rows_pp); one(r + 2LL * rows_pp); one(r + 3LL * rows_pp); }
            for (; r < r1; r += rows_pp) one(r);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[tid][k] = s0[k]; red[tid][4 + k] = s1[k]; }
        __syncthreads();
        if (rsub == 0 && c < C) {
            float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int u = 0; u < rows_pp; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] += red[u * cg + cgi][k];
            float* dst = part + ((long long)(grp * chunks + chunk) * 2) * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[c + k] = t[k]; dst[C + c + k] = t[4 + k]; }
        }
        __syncthreads();
    }
}

// FIN 0: mean / rstd (+ optional running stats, BatchNorm semantics)   FIN 1: raw sums -> out0/out1
// block (64, 4): x = output index, y = lane over the chunk list (fixed order: lane sums then y = 0..3)
template <int FIN>
__global__ void stat_final_kernel(const float* __restrict__ part, int chunks, int C, int groups, long long P, float eps,
                                  float momentum, float* __restrict__ out0, float* __restrict__ out1,
                                  float* __restrict__ run_mean, float* __restrict__ run_var) {
    // block (64, SY <= 16): lane y sums chunks y, y + SY, ... (two loads pairs in flight), the SY partials meet in LDS in lane
    // order.  With 4 lanes the 2 M-row BatchNorm layers (1000+ chunks) spent 70 us here on a serial chain of loads.
    __shared__ double red[2][16][64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int SY = blockDim.y;
    const bool ok = i < groups * C;
    const int grp = ok ? i / C : 0, c = ok ? i - grp * C : 0;
    double s0 = 0.0, s1 = 0.0;
    if (ok) {
        int k = threadIdx.y;
        for (; k + SY < chunks; k += 2 * SY) {
            const float* a = part + ((long long)(grp * chunks + k) * 2) * C;
            const float* b = part + ((long long)(grp * chunks + k + SY) * 2) * C;
            const float a0 = a[c], a1 = a[C + c], b0 = b[c], b1 = b[C + c];
            s0 += (double)a0; s1 += (double)a1; s0 += (double)b0; s1 += (double)b1;
        }
        for (; k < chunks; k += SY) {
            const float* src = part + ((long long)(grp * chunks + k) * 2) * C;
            s0 += (double)src[c]; s1 += (double)src[C + c];
        }
    }
    red[0][threadIdx.y][threadIdx.x] = s0; red[1][threadIdx.y][threadIdx.x] = s1;
    __syncthreads();
    if (!ok || threadIdx.y != 0) return;
    s0 = 0.0; s1 = 0.0;
    for (int y = 0; y < SY; ++y) { s0 += red[0][y][threadIdx.x]; s1 += red[1][y][threadIdx.x]; }
    if (FIN == 0) {
        const double m = s0 / (double)P;
        double var = s1 / (double)P - m * m; if (var < 0.0) var = 0.0;
        out0[i] = (float)m;
        out1[i] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {   // nn.BatchNorm2d: running_var uses the unbiased estimate
            const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
            run_mean[i] = (float)((1.0 - momentum) * (double)run_mean[i] + momentum * m);
            run_var[i] = (float)((1.0 - momentum) * (double)run_var[i] + momentum * unb);
        }
    } else { out0[i] = (float)s0; out1[i] = (float)s1; }
}

static int stat_final_lanes(int chunks) { int y = 4; while (y < 16 && y * 8 < chunks) y <<= 1; return y; }

template <int MODE, typename T>
static int launch_stats(const T* a, int lda, const T* b, int ldb, const T* g, int ldg, const float* mean,
                        const float* rstd, int stat_per_group, int groups, long long P, int C, float* part, hipStream_t s, int plan_groups = 0,
                        const T* xlo = nullptr, int ldxlo = 0, int Wlo = 0) {
    // plan_groups: take the chunking of a launch with that many groups (grouped BatchNorm chunks each group exactly as a call of its own would)
    const StatPlan sp = stat_plan(plan_groups > 0 ? plan_groups : groups, P);
    const bool vec = vec4_ok(a, lda, C) && (MODE == 0 || (xlo ? vec4_ok(xlo, ldxlo, C) : vec4_ok(b, ldb, C))) && (MODE != 2 || vec4_ok(g, ldg, C)) &&
                     ((C >> 2) >= 256 ? (C >> 2) % 256 == 0 : 256 % (C >> 2) == 0);
    if (xlo && !vec) return MRDIS_EUNSUPPORTED;
    if (vec && xlo) {
        if constexpr (MODE == 2)                          // (its own instantiation: the interpolation's registers do not ride on the plain statistics passes)
            MRDIS_LAUNCH((stat_partial_vec_kernel<2, T, true>), dim3(sp.chunks, groups), dim3(256), 0, s, a, lda, b, ldb, g, ldg, mean, rstd,
                               stat_per_group, P, C, sp.rpb, part, xlo, ldxlo, Wlo);
        else return MRDIS_EINVAL;
    } else if (vec)
        MRDIS_LAUNCH((stat_partial_vec_kernel<MODE, T>), dim3(sp.chunks, groups), dim3(256), 0, s, a, lda, b, ldb, g, ldg, mean, rstd,
                           stat_per_group, P, C, sp.rpb, part, nullptr, 0, 0);
    else
    MRDIS_LAUNCH((stat_partial_kernel<MODE, T>), dim3(sp.chunks, groups), dim3(64, 4), 0, s, a, lda, b, ldb, g, ldg, mean, rstd,
                       stat_per_group, P, C, sp.rpb, part);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ BatchNorm (training)
template <int V, typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                long long P, int C, long long Pg) {
    // Pg: rows per statistics group (P for plain BatchNorm; P / G when the batch holds G groups that the reference normalises in G
    // separate calls -- the modalities of one encoder pass): row r uses mean / rstd [(r / Pg) * C + c]
    const int Q = C / V;
    EW_LOOP(P * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        const int so = (int)(r / Pg) * C + c;
        Vec<V> a, o; a.load(x + r * ldx + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float sc = rstd[so + k] * (gamma ? gamma[c + k] : 1.f);
            o.v[k] = (a.v[k] - mean[so + k]) * sc + (beta ? beta[c + k] : 0.f);
        }
        o.store(y + r * ldy + c);
    }
}

// the statistics of G groups of one BatchNorm layer in one launch: per group mean / rstd, and the running statistics updated group by group
// in order (nn.BatchNorm2d called G times: running = (1 - momentum) running + momentum stat_g, g = 0 .. G - 1).  block (64, SY).
__global__ void stat_final_groups_kernel(const float* __restrict__ part, int chunks, int C, int groups, long long P, float eps, float momentum,
                                         float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ run_mean, float* __restrict__ run_var) {
    __shared__ double red[2][16][64];
    const int c = blockIdx.x * 64 + threadIdx.x, SY = blockDim.y;
    const bool ok = c < C;
    double rm = 0.0, rv = 0.0;
    if (ok && run_mean && threadIdx.y == 0) { rm = (double)run_mean[c]; rv = (double)run_var[c]; }
    for (int g = 0; g < groups; ++g) {
        double s0 = 0.0, s1 = 0.0;
        if (ok)
            for (int k = threadIdx.y; k < chunks; k += SY) {
                const float* src = part + ((long long)(g * chunks + k) * 2) * C;
                s0 += (double)src[c]; s1 += (double)src[C + c];
            }
        red[0][threadIdx.y][threadIdx.x] = s0; red[1][threadIdx.y][threadIdx.x] = s1;
        __syncthreads();
        if (ok && threadIdx.y == 0) {
            s0 = 0.0; s1 = 0.0;
            for (int y = 0; y < SY; ++y) { s0 += red[0][y][threadIdx.x]; s1 += red[1][y][threadIdx.x]; }
            const double m = s0 / (double)P;
            double var = s1 / (double)P - m * m; if (var < 0.0) var = 0.0;
            out0[g * C + c] = (float)m;
            out1[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
            if (run_mean) {
                const double unb = P > 1 ? var * (double)P / (double)(P - 1) : var;
                // each call of the reference rounds the running statistics to fp32: round between the groups as well
                rm = (double)(float)((1.0 - momentum) * rm + momentum * m);
                rv = (double)(float)((1.0 - momentum) * rv + momentum * unb);
            }
        }
        __syncthreads();
    }
    if (ok && run_mean && threadIdx.y == 0) { run_mean[c] = (float)rm; run_var[c] = (float)rv; }
}

template <typename T>
static int bn_train_fwd_impl(const T* x, int ldx, T* y, int ldy, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var,
                                  float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                  long long P, int C, float eps, float momentum, int G, void* stream) {
    // G groups of P rows each (G = 1: plain BatchNorm over P rows); save_mean / save_rstd hold G * C entries
    if (!x || !y || !save_mean || !save_rstd || !workspace || P < 1 || C < 1 || ldx < C || ldy < C || G < 1 || G > 64) return MRDIS_EINVAL;
    // every group is chunked as a call of its own would be (the same partial sums in the same order: grouped == G separate calls, bit for bit)
    if (workspace_bytes < (size_t)G * mrdis_norm_workspace(1, P, C)) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    int rc = launch_stats<0, T>(x, ldx, (const T*)nullptr, 0, (const T*)nullptr, 0, nullptr, nullptr, 0, G, P, C, part, s, 1);
    if (rc) return rc;
    const StatPlan sp = stat_plan(1, P);
    if (G == 1)
        MRDIS_LAUNCH((stat_final_kernel<0>), dim3(mrdis_cdiv(C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, 1, P, eps, momentum,
                           save_mean, save_rstd, running_mean, running_mean ? running_var : nullptr);
    else
        MRDIS_LAUNCH(stat_final_groups_kernel, dim3(mrdis_cdiv(C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, G, P, eps, momentum,
                           save_mean, save_rstd, running_mean, running_mean ? running_var : nullptr);
    MRDIS_CHECK_LAUNCH();
    const long long rows = P * G;
    if (vec4_ok(x, ldx, C) && vec4_ok(y, ldy, C))
        MRDIS_LAUNCH((bn_apply_kernel<4, T>), dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, save_mean, save_rstd, rows, C, P);
    else
        MRDIS_LAUNCH((bn_apply_kernel<1, T>), dim3(ew_blocks(rows * C)), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, save_mean, save_rstd, rows, C, P);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// inference-mode BatchNorm (evaluate(), main_missing.py:338): per-channel affine from the running statistics
template <int V, typename T>
__global__ void bn_eval_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ rmean, const float* __restrict__ rvar,
                               float eps, long long P, int C) {
    const int Q = C / V;
    EW_LOOP(P * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        Vec<V> a, o; a.load(x + r * ldx + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float sc = rsqrtf(rvar[c + k] + eps) * (gamma ? gamma[c + k] : 1.f);
            o.v[k] = (a.v[k] - rmean[c + k]) * sc + (beta ? beta[c + k] : 0.f);
        }
        o.store(y + r * ldy + c);
    }
}
template <typename T>
static int bn_eval_fwd_impl(const T* x, int ldx, T* y, int ldy, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, long long P, int C, float eps, void* stream) {
    if (!x || !y || !running_mean || !running_var || P < 1 || C < 1 || ldx < C || ldy < C) return MRDIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (vec4_ok(x, ldx, C) && vec4_ok(y, ldy, C))
        MRDIS_LAUNCH((bn_eval_kernel<4, T>), dim3(ew_blocks(P * C / 4)), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, running_mean, running_var, eps, P, C);
    else
        MRDIS_LAUNCH((bn_eval_kernel<1, T>), dim3(ew_blocks(P * C)), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta, running_mean, running_var, eps, P, C);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

template <int V, typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ sdy, const float* __restrict__ sdyxh, T* __restrict__ dx, int lddx,
                                    long long P, int C, float* __restrict__ acc_dgamma, float* __restrict__ acc_dbeta, long long Pg, int G) {
    // Pg rows per statistics group, G groups (P = G * Pg rows): statistics and sums are indexed [group][channel]
    const int Q = C / V;
    const float invP = 1.f / (float)Pg;
    if (acc_dgamma != nullptr && blockIdx.x == 0)          // fold this call's parameter gradients into the running sums (groups in order)
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            for (int g = 0; g < G; ++g) { acc_dgamma[c] += sdyxh[g * C + c]; acc_dbeta[c] += sdy[g * C + c]; }
    EW_LOOP(P * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        const int so = (int)(r / Pg) * C + c;
        Vec<V> d, a, o; d.load(dy + r * lddy + c); a.load(x + r * ldx + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float rs = rstd[so + k];
            const float xh = (a.v[k] - mean[so + k]) * rs;
            o.v[k] = (gamma ? gamma[c + k] : 1.f) * rs * (d.v[k] - sdy[so + k] * invP - xh * sdyxh[so + k] * invP);
        }
        o.store(dx + r * lddx + c);
    }
}

template <typename T>
static int bn_train_bwd_impl(const T* dy, int lddy, const T* x, int ldx, const float* gamma,
                                  const float* save_mean, const float* save_rstd, T* dx, int lddx,
                                  float* dgamma, float* dbeta, float* acc_dgamma, float* acc_dbeta,
                                  void* workspace, size_t workspace_bytes, long long P, int C, int G, void* stream) {
    // G groups of P rows (see bn_train_fwd_impl); dgamma / dbeta receive G * C per-group sums
    if (!dy || !x || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !workspace || P < 1 || C < 1 || G < 1 || G > 64) return MRDIS_EINVAL;
    if ((acc_dgamma == nullptr) != (acc_dbeta == nullptr)) return MRDIS_EINVAL;
    if (workspace_bytes < (size_t)G * mrdis_norm_workspace(1, P, C)) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    int rc = launch_stats<1, T>(dy, lddy, x, ldx, (const T*)nullptr, 0, save_mean, save_rstd, 1, G, P, C, part, s, 1);
    if (rc) return rc;
    const StatPlan sp = stat_plan(1, P);
    // dbeta = sum dy ; dgamma = sum dy * xhat  (per group)
    MRDIS_LAUNCH((stat_final_kernel<1>), dim3(mrdis_cdiv(G * C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, G, P, 0.f, 0.f,
                       dbeta, dgamma, nullptr, nullptr);
    MRDIS_CHECK_LAUNCH();
    const long long rows = P * G;
    if (vec4_ok(dy, lddy, C) && vec4_ok(x, ldx, C) && vec4_ok(dx, lddx, C))
        MRDIS_LAUNCH((bn_bwd_apply_kernel<4, T>), dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, dy, lddy, x, ldx, gamma, save_mean, save_rstd, dbeta, dgamma, dx, lddx, rows, C, acc_dgamma, acc_dbeta, P, G);
    else
        MRDIS_LAUNCH((bn_bwd_apply_kernel<1, T>), dim3(ew_blocks(rows * C)), dim3(256), 0, s, dy, lddy, x, ldx, gamma, save_mean, save_rstd, dbeta, dgamma, dx, lddx, rows, C, acc_dgamma, acc_dbeta, P, G);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ InstanceNorm + SPADE modulation
template <int V, typename T>
__global__ void spade_fwd_kernel(const T* __restrict__ z, int ldz, const T* __restrict__ g, int ldg, const T* __restrict__ b, int ldb,
                                 T* __restrict__ out, int ldo, const float* __restrict__ mean, const float* __restrict__ rstd,
                                 long long HW, long long rows, int C) {
    const int Q = C / V;
    EW_LOOP(rows * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        const int n = (int)(r / HW);
        Vec<V> a, gg, bb, o; a.load(z + r * ldz + c); gg.load(g + r * ldg + c); bb.load(b + r * ldb + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float zh = (a.v[k] - mean[n * C + c + k]) * rstd[n * C + c + k];
            o.v[k] = zh * (1.f + gg.v[k]) + bb.v[k];
        }
        o.store(out + r * ldo + c);
    }
}

template <typename T>
static int instnorm_spade_fwd_impl(const T* z, int ldz, const T* gamma, int ldg,
                                        const T* beta, int ldb, T* out, int ldo,
                                        float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                        int N, long long HW, int C, float eps, void* stream) {
    if (!z || !gamma || !beta || !out || !save_mean || !save_rstd || N < 1 || HW < 1 || C < 1) return MRDIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (workspace != nullptr) {        // workspace == NULL: save_mean / save_rstd already hold the statistics of z (e.g. from mrdis_bilinear_up2_stats_fwd)
        if (workspace_bytes < mrdis_norm_workspace(N, HW, C)) return MRDIS_EWORKSPACE;
        float* part = reinterpret_cast<float*>(workspace);
        int rc = launch_stats<0, T>(z, ldz, (const T*)nullptr, 0, (const T*)nullptr, 0, nullptr, nullptr, 0, N, HW, C, part, s);
        if (rc) return rc;
        const StatPlan sp = stat_plan(N, HW);
        MRDIS_LAUNCH((stat_final_kernel<0>), dim3(mrdis_cdiv(N * C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, N, HW, eps, 0.f,
                           save_mean, save_rstd, nullptr, nullptr);
        MRDIS_CHECK_LAUNCH();
    }
    const long long rows = (long long)N * HW;
    if (vec4_ok(z, ldz, C) && vec4_ok(gamma, ldg, C) && vec4_ok(beta, ldb, C) && vec4_ok(out, ldo, C))
        MRDIS_LAUNCH((spade_fwd_kernel<4, T>), dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, z, ldz, gamma, ldg, beta, ldb, out, ldo, save_mean, save_rstd, HW, rows, C);
    else
        MRDIS_LAUNCH((spade_fwd_kernel<1, T>), dim3(ew_blocks(rows * C)), dim3(256), 0, s, z, ldz, gamma, ldg, beta, ldb, out, ldo, save_mean, save_rstd, HW, rows, C);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

template <int V, typename T>
__global__ void spade_bwd_kernel(const T* __restrict__ dout, int lddo, const T* __restrict__ z, int ldz, const T* __restrict__ g, int ldg,
                                 const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ s0,
                                 const float* __restrict__ s1, T* __restrict__ dz, int lddz, T* __restrict__ dgm, int lddg,
                                 T* __restrict__ dbt, int lddb, long long HW, long long rows, int C) {
    const int Q = C / V;
    const float inv = 1.f / (float)HW;
    EW_LOOP(rows * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        const int n = (int)(r / HW);
        Vec<V> d, a, gg, oz, og; d.load(dout + r * lddo + c); a.load(z + r * ldz + c); gg.load(g + r * ldg + c);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int si = n * C + c + k;
            const float rs = rstd[si];
            const float zh = (a.v[k] - mean[si]) * rs;
            const float dzh = d.v[k] * (1.f + gg.v[k]);
            og.v[k] = d.v[k] * zh;                                            // d gamma
            oz.v[k] = rs * (dzh - s0[si] * inv - zh * s1[si] * inv);          // instance-norm backward
        }
        oz.store(dz + r * lddz + c); og.store(dgm + r * lddg + c);
        if (dbt) d.store(dbt + r * lddb + c);                                 // d beta = dout
    }
}

template <typename T>
static int instnorm_spade_bwd_impl(const T* dout, int lddo, const T* z, int ldz,
                                        const T* gamma, int ldg, const float* save_mean, const float* save_rstd,
                                        T* dz, int lddz, T* dgamma, int lddg, T* dbeta, int lddb,
                                        void* workspace, size_t workspace_bytes,
                                        int N, long long HW, int C, void* stream) {
    if (!dout || !z || !gamma || !save_mean || !save_rstd || !dz || !dgamma || !workspace || N < 1 || HW < 1 || C < 1) return MRDIS_EINVAL;
    // layout of workspace: partials, then the two (N*C) reduced sums
    const size_t pbytes = mrdis_norm_workspace(N, HW, C);
    const size_t need = pbytes + sizeof(float) * 2 * (size_t)N * C;
    if (workspace_bytes < need) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    float* s0 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pbytes);
    float* s1 = s0 + (size_t)N * C;
    int rc = launch_stats<2, T>(dout, lddo, z, ldz, gamma, ldg, save_mean, save_rstd, 1, N, HW, C, part, s);
    if (rc) return rc;
    const StatPlan sp = stat_plan(N, HW);
    MRDIS_LAUNCH((stat_final_kernel<1>), dim3(mrdis_cdiv(N * C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, N, HW, 0.f, 0.f, s0, s1, nullptr, nullptr);
    MRDIS_CHECK_LAUNCH();
    const long long rows = (long long)N * HW;
    const bool v = vec4_ok(dout, lddo, C) && vec4_ok(z, ldz, C) && vec4_ok(gamma, ldg, C) && vec4_ok(dz, lddz, C) &&
                   vec4_ok(dgamma, lddg, C) && (!dbeta || vec4_ok(dbeta, lddb, C));
    if (v)
        MRDIS_LAUNCH((spade_bwd_kernel<4, T>), dim3(ew_blocks(rows * C / 4)), dim3(256), 0, s, dout, lddo, z, ldz, gamma, ldg, save_mean, save_rstd, s0, s1, dz, lddz, dgamma, lddg, dbeta, lddb, HW, rows, C);
    else
        MRDIS_LAUNCH((spade_bwd_kernel<1, T>), dim3(ew_blocks(rows * C)), dim3(256), 0, s, dout, lddo, z, ldz, gamma, ldg, save_mean, save_rstd, s0, s1, dz, lddz, dgamma, lddg, dbeta, lddb, HW, rows, C);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
// SPADE backward whose z is the result of nn.Upsample(scale_factor=2, bilinear) (model.py:2551-2573 in front of :2440-2446): the apply pass of
// instnorm_spade_bwd and the adjoint of the resize (bilinear_up2_bwd_kernel) in one kernel.  A workgroup owns 8 x 8 pixels of the LOW-resolution map x
// 32 channels: phase 1 forms dz at the 18 x 18 full-resolution pixels that read them (each once: 3 streaming loads, d gamma / d beta stored by the
// workgroup whose 16 x 16 block holds the pixel) into LDS, phase 2 sums the 4 x 4 neighbourhood of every low-resolution pixel (rows 2i-1 .. 2i+2, weights
// 0.25 0.75 0.75 0.25; at a border the clamped row / column) from LDS -- the full-resolution dz (N x 2H x 2W x C, 268 MB on the last level at B = 32)
// is neither written nor read back; the one-pixel halo is read by two workgroups (1.27 x the loads, the second mostly from L2).  Same expressions as
// spade_bwd_kernel and bilinear_up2_bwd_kernel in the same order.  (A plain gather -- every thread forming the 16 dz of its pixel itself -- was 2 x slower
// than the two kernels it replaces: 48 dependent-latency loads per output.)
// ONEPASS (fp32, xlo given): no statistics pass at all.  d z = rstd (dzh - s0 / HW - zh s1 / HW) is linear in its three terms and so is the resize's adjoint U^T:
//   d x = rstd (U^T dzh - 4 s0 / HW - (s1 / HW) rstd (U^T U x - 4 mean)),      dzh = dout (1 + gamma), U^T 1 = 4, U^T zh = rstd (U^T z - 4 mean), z = U x
// so this kernel writes A = U^T dzh into dx and the partial sums of (dzh, dzh zh) over its own 16 x 16 pixels (one `part` chunk per tile, the layout
// stat_final_kernel<1> reads), and spade_bwd_up2_final_kernel finishes d x in place from A, the sums and the 3 x 3 stencil U^T U of the LOW-resolution x:
// dout and gamma are read once instead of twice, z never (0.7 GB less per full-resolution block at B = 32).
constexpr int UB_T = 8, UB_R = 2 * UB_T + 2, UB_CC = 32;
// NT threads per workgroup (256 | 512): the two LDS tiles (54 KB) allow two workgroups per CU either way, so 512 threads double the waves that hide the
// latency of phase 1's streaming loads; those loads (dout, gamma of every pixel a thread visits: <= UB_IT<NT> pixels) are ALL issued before the first is
// used (coordinates clamped, a flag instead of a branch), so a thread has 2 x UB_IT 16-byte (fp32) loads in flight instead of 2.
template <int NT> struct UB_IT { static constexpr int value = (UB_R * UB_R + NT / 8 - 1) / (NT / 8); };      // pixels per thread at 8 channel quads (fewer quads: fewer pixels)
template <typename T, bool ONEPASS = false, int NT = 256, bool XLO = true>
__global__ __launch_bounds__(NT) void spade_bwd_up2_kernel(const T* __restrict__ dout, int lddo, const T* __restrict__ z, int ldz, const T* __restrict__ g, int ldg,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ s0,
                                                            const float* __restrict__ s1, T* __restrict__ dxl, int lddx, T* __restrict__ dgm, int lddg,
                                                            T* __restrict__ dbt, int lddb, int Hi, int Wi, int C, int tiles_x,
                                                            const T* __restrict__ xlo, int ldxlo, float* __restrict__ part = nullptr,
                                                            float* __restrict__ abuf = nullptr, int abl = 0) {      // abuf (ONEPASS, bf16 maps): A leaves in fp32, dense (N, Hi, Wi, C); abl: timing-only ablations (-DSPADE_UP2_ABL builds)
#ifndef SPADE_UP2_ABL
    abl = 0;
#endif
    // xlo: z is not stored -- it is the x2 resize of xlo (N, Hi, Wi, C); the workgroup's 10 x 10 low-resolution neighbourhood goes through LDS and every
    // full-resolution z is interpolated from it exactly as bilinear_up2_fwd_kernel formed (and stored) it
    __shared__ __attribute__((aligned(16))) float tile[UB_R * UB_R * UB_CC];         // dz of the 18 x 18 pixels x 32 channels: 41.5 KB
    __shared__ __attribute__((aligned(16))) float xt[(UB_T + 2) * (UB_T + 2) * UB_CC];   // low-resolution rows i0-1 .. i0+8, columns j0-1 .. j0+8 (clamped): 12.8 KB
    const int tix = blockIdx.x % tiles_x, tiy = blockIdx.x / tiles_x, n = blockIdx.y;
    const int c0 = UB_CC * blockIdx.z, cw = (C - c0 < UB_CC ? C - c0 : UB_CC), Q = cw / 4;
    const int i0 = UB_T * tiy, j0 = UB_T * tix;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const float inv = 1.f / (float)((long long)Ho * Wo);
    const long long img = (long long)n * Ho * Wo;
    const int tid = threadIdx.x;
    // ---- phase 1: dz (and d gamma, d beta) of the full-resolution pixels (2 i0 - 1 + ry, 2 j0 - 1 + rx), ry, rx < 18
    // (index arithmetic: a timing-only build without any memory traffic still took 40 % of the kernel's time, and a third of its VALU stream was address
    //  arithmetic -- 85 v_mul_lo_u32 and 35 v_mad_u64_u32 at a quarter of the rate.  Q is a power of two (host), the pixel walk is incremental, and a pixel's
    //  offset inside its image is a 24-bit multiply on top of per-image base pointers)
    const int lq = 31 - __clz(Q);                                                     // host: Q in {1, 2, 4, 8}
    const int q1 = tid & (Q - 1), p1 = tid >> lq, pstep = NT >> lq;
    const int dry = pstep / UB_R, drx = pstep - dry * UB_R;                           // (wave-uniform)
    constexpr int MAXIT = UB_IT<NT>::value;
    const int c = c0 + 4 * q1;
    const T* const dout_n = dout + img * lddo + c;
    const T* const g_n = g + img * ldg + c;
    const T* const z_n = XLO ? nullptr : z + img * ldz + c;
    T* const dgm_n = dgm + img * lddg + c;
    T* const dbt_n = dbt ? dbt + img * lddb + c : nullptr;
    // all streaming loads of this thread first (a pixel outside the image / past the tile: the image's pixel 0, flagged off)
    Vec<4> dv[MAXIT], gv[MAXIT], zv[XLO ? 1 : MAXIT];
    int pixv[MAXIT];                                                                  // pixel index inside the image, or -1
    {
        int ry = p1 / UB_R, rx = p1 - (p1 / UB_R) * UB_R, px = p1;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int h = 2 * i0 - 1 + ry, w_ = 2 * j0 - 1 + rx;
            const bool ok = px < UB_R * UB_R && (unsigned)h < (unsigned)Ho && (unsigned)w_ < (unsigned)Wo;
            pixv[it] = ok ? (int)__umul24((unsigned)h, (unsigned)Wo) + w_ : -1;
            const unsigned pl = ok ? (unsigned)pixv[it] : 0u;
            if (!(abl & 1)) { dv[it].load(dout_n + __umul24(pl, (unsigned)lddo)); gv[it].load(g_n + __umul24(pl, (unsigned)ldg)); }
            else { for (int k = 0; k < 4; ++k) { dv[it].v[k] = 1.f + k; gv[it].v[k] = 0.5f; } }
            if (!XLO) zv[it].load(z_n + __umul24(pl, (unsigned)ldz));
            px += pstep; rx += drx; ry += dry;
            if (rx >= UB_R) { rx -= UB_R; ry += 1; }
        }
    }
    float mu[4], rs[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int si = n * C + c0 + 4 * q1 + k; mu[k] = mean[si]; rs[k] = rstd[si];
        a0[k] = ONEPASS ? 0.f : s0[si] * inv; a1[k] = ONEPASS ? 0.f : s1[si] * inv;
    }
    float ps0[4] = {0.f, 0.f, 0.f, 0.f}, ps1[4] = {0.f, 0.f, 0.f, 0.f};
    if (XLO) {
        for (int px = p1; px < (UB_T + 2) * (UB_T + 2); px += pstep) {
            const int ly = px / (UB_T + 2), lx = px - ly * (UB_T + 2);
            int i = i0 - 1 + ly, j = j0 - 1 + lx;
            i = i < 0 ? 0 : (i > Hi - 1 ? Hi - 1 : i); j = j < 0 ? 0 : (j > Wi - 1 ? Wi - 1 : j);
            *reinterpret_cast<float4*>(xt + (px * (UB_CC / 4) + q1) * 4) = ld4(xlo + (((long long)n * Hi + i) * Wi + j) * ldxlo + c0 + 4 * q1);
        }
        __syncthreads();
    }
    int ry_ = p1 / UB_R, rx_ = p1 - (p1 / UB_R) * UB_R, px_ = p1;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int px = px_, ry = ry_, rx = rx_;
        px_ += pstep; rx_ += drx; ry_ += dry;
        if (rx_ >= UB_R) { rx_ -= UB_R; ry_ += 1; }
        if (pixv[it] < 0) continue;
        const int h = 2 * i0 - 1 + ry, w_ = 2 * j0 - 1 + rx;
        const unsigned pix = (unsigned)pixv[it];
        const Vec<4>& d = dv[it]; const Vec<4>& gg = gv[it];
        Vec<4> zz, og;
        if (XLO && (abl & 2)) { for (int k = 0; k < 4; ++k) zz.v[k] = 0.25f * k; }
        else if (XLO) {
            // (h, w) -> low-resolution rows / columns and weights as up2_value; tile coordinates = low-resolution index - (i0 - 1), clamped rows are copies
            const int i = h >> 1, j = w_ >> 1;
            int r0, r1, q0_, q1_; float A0, A1, B0, B1;
            if (h & 1) { r0 = i; r1 = i < Hi - 1 ? i + 1 : Hi - 1; A0 = 0.75f; A1 = 0.25f; }
            else { r0 = i > 0 ? i - 1 : 0; r1 = i; A0 = i > 0 ? 0.25f : 0.f; A1 = i > 0 ? 0.75f : 1.f; }
            if (w_ & 1) { q0_ = j; q1_ = j < Wi - 1 ? j + 1 : Wi - 1; B0 = 0.75f; B1 = 0.25f; }
            else { q0_ = j > 0 ? j - 1 : 0; q1_ = j; B0 = j > 0 ? 0.25f : 0.f; B1 = j > 0 ? 0.75f : 1.f; }
            r0 -= i0 - 1; r1 -= i0 - 1; q0_ -= j0 - 1; q1_ -= j0 - 1;
            const float4 p00 = *reinterpret_cast<const float4*>(xt + ((r0 * (UB_T + 2) + q0_) * (UB_CC / 4) + q1) * 4);
            const float4 p01 = *reinterpret_cast<const float4*>(xt + ((r0 * (UB_T + 2) + q1_) * (UB_CC / 4) + q1) * 4);
            const float4 p10 = *reinterpret_cast<const float4*>(xt + ((r1 * (UB_T + 2) + q0_) * (UB_CC / 4) + q1) * 4);
            const float4 p11 = *reinterpret_cast<const float4*>(xt + ((r1 * (UB_T + 2) + q1_) * (UB_CC / 4) + q1) * 4);
            const float* a_ = &p00.x; const float* b_ = &p01.x; const float* c_ = &p10.x; const float* d_ = &p11.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) zz.v[k] = (float)(T)(A0 * (B0 * a_[k] + B1 * b_[k]) + A1 * (B0 * c_[k] + B1 * d_[k]));
        } else zz = zv[XLO ? 0 : it];
        float4 dz;
        float* dzp = &dz.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float zh = (zz.v[k] - mu[k]) * rs[k];
            const float dzh = d.v[k] * (1.f + gg.v[k]);
            og.v[k] = d.v[k] * zh;                                                    // d gamma
            dzp[k] = ONEPASS ? dzh : rs[k] * (dzh - a0[k] - zh * a1[k]);              // instance-norm backward (ONEPASS: its first term only)
        }
        *reinterpret_cast<float4*>(tile + (px * (UB_CC / 4) + q1) * 4) = dz;
        if (ry >= 1 && ry <= 2 * UB_T && rx >= 1 && rx <= 2 * UB_T) {                 // this workgroup's own 16 x 16 pixels
            if (!(abl & 4)) {
            og.store(dgm_n + __umul24(pix, (unsigned)lddg));
            if (dbt) d.store(dbt_n + __umul24(pix, (unsigned)lddb));                  // d beta = dout
            }
            if (ONEPASS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { ps0[k] += dzp[k]; ps1[k] += dzp[k] * ((zz.v[k] - mu[k]) * rs[k]); }
            }
        }
    }
    if (ONEPASS) {
        // the 64 / Q lanes of a wave that share a channel quad (lane % Q: Q divides 64) are added by a butterfly of fixed shape, each wave writes its own `part` chunk
        // (chunk = (NT / 64) tile + wave): no LDS round trip, no extra barrier; bit-reproducible
        const int lane = tid & 63, wv = tid >> 6;
        for (int o = Q; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { ps0[k] += __shfl_xor(ps0[k], o, 64); ps1[k] += __shfl_xor(ps1[k], o, 64); }
        }
        if (lane < Q) {
            float* dst = part + ((long long)((n * (int)gridDim.x + (int)blockIdx.x) * (NT / 64) + wv) * 2) * C + c0;      // group = image n, chunk = (NT / 64) tile + wave
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[4 * lane + k] = ps0[k]; dst[C + 4 * lane + k] = ps1[k]; }
        }
    }
    __syncthreads();
    // ---- phase 2: the resize's adjoint from LDS
    if (abl & 8) return;
    for (int it = tid; it < UB_T * UB_T * Q; it += NT) {
        const int q = it & (Q - 1), lp = it >> lq, li = lp / UB_T, lj = lp - li * UB_T;
        const int i = i0 + li, j = j0 + lj;
        if (i >= Hi || j >= Wi) continue;
        int rr[4] = {2 * li, 2 * li + 1, 2 * li + 2, 2 * li + 3};                     // tile rows of full-resolution rows 2i-1 .. 2i+2
        int cc[4] = {2 * lj, 2 * lj + 1, 2 * lj + 2, 2 * lj + 3};
        if (i == 0) rr[0] = 1;                                                        // out[0] = 1.0 x[0]: the missing 0.25 comes from row 0 itself
        if (i == Hi - 1) rr[3] = rr[2];                                               // out[Ho-1] = 1.0 x[Hi-1]
        if (j == 0) cc[0] = 1;
        if (j == Wi - 1) cc[3] = cc[2];
        const float wr[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        Vec<4> acc;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc.v[k] = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) t[b] = *reinterpret_cast<const float4*>(tile + ((rr[a] * UB_R + cc[b]) * (UB_CC / 4) + q) * 4);
            const float* t0 = &t[0].x; const float* t1 = &t[1].x; const float* t2 = &t[2].x; const float* t3 = &t[3].x;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc.v[k] += wr[a] * ((0.25f * t0[k] + 0.75f * t1[k]) + (0.75f * t2[k] + 0.25f * t3[k]));
        }
        if (ONEPASS && abuf != nullptr) *reinterpret_cast<float4*>(abuf + (((long long)n * Hi + i) * Wi + j) * C + c0 + 4 * q) = make_float4(acc.v[0], acc.v[1], acc.v[2], acc.v[3]);
        else acc.store(dxl + (((long long)n * Hi + i) * Wi + j) * lddx + c0 + 4 * q);
    }
}

// d x = rstd (A - 4 s0 / HW - (s1 / HW) rstd (U^T U x - 4 mean)) in place on dx (which holds A = U^T dzh); U^T U along one axis is the 3-tap stencil that the x2
// resize followed by its adjoint makes of x -- (0.375, 1.25, 0.375) inside, built here from the same clamped rows and weights as the two kernels (so every border
// case, down to a one-pixel axis, is the composition of what they do)
__device__ __forceinline__ void up2_gram_axis(int i, int Hi, float c[3], int idx[3]) {
    const int Ho = 2 * Hi;
    int rr[4] = {2 * i - 1, 2 * i, 2 * i + 1, 2 * i + 2};
    const float wr[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    if (i == 0) rr[0] = 0;
    if (i == Hi - 1) rr[3] = Ho - 1;
    c[0] = c[1] = c[2] = 0.f;
    idx[0] = i > 0 ? i - 1 : 0; idx[1] = i; idx[2] = i < Hi - 1 ? i + 1 : Hi - 1;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int h = rr[a], ii = h >> 1;
        int r0, r1; float A0, A1;
        if (h & 1) { r0 = ii; r1 = ii < Hi - 1 ? ii + 1 : Hi - 1; A0 = 0.75f; A1 = 0.25f; }
        else { r0 = ii > 0 ? ii - 1 : 0; r1 = ii; A0 = ii > 0 ? 0.25f : 0.f; A1 = ii > 0 ? 0.75f : 1.f; }
        c[r0 - i + 1] += wr[a] * A0; c[r1 - i + 1] += wr[a] * A1;                    // r0, r1 in {i - 1, i, i + 1}
    }
}
template <typename T>
__global__ void spade_bwd_up2_final_kernel(T* __restrict__ dx, int lddx, const T* __restrict__ xlo, int ldxlo, const float* __restrict__ mean,
                                           const float* __restrict__ rstd, const float* __restrict__ s0, const float* __restrict__ s1,
                                           int N, int Hi, int Wi, int C, const float* __restrict__ abuf) {
    const int Q = C / 4;
    const float inv = 1.f / (4.f * (float)Hi * (float)Wi);
    EW_LOOP((long long)N * Hi * Wi * Q) {
        const long long pix = idx / Q; const int c = (int)(idx - pix * Q) * 4;
        const int j = (int)(pix % Wi); const long long t = pix / Wi; const int i = (int)(t % Hi), n = (int)(t / Hi);
        float cr[3], cc[3]; int ir[3], ic[3];
        up2_gram_axis(i, Hi, cr, ir); up2_gram_axis(j, Wi, cc, ic);
        float Tg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float rowv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float4 v = ld4(xlo + (((long long)n * Hi + ir[a]) * Wi + ic[b]) * ldxlo + c);
                rowv[0] += cc[b] * v.x; rowv[1] += cc[b] * v.y; rowv[2] += cc[b] * v.z; rowv[3] += cc[b] * v.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) Tg[k] += cr[a] * rowv[k];
        }
        Vec<4> A, o;
        if (abuf != nullptr) { const float4 av = *reinterpret_cast<const float4*>(abuf + pix * C + c); A.v[0] = av.x; A.v[1] = av.y; A.v[2] = av.z; A.v[3] = av.w; }
        else A.load(dx + pix * lddx + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int si = n * C + c + k;
            const float rs = rstd[si];
            o.v[k] = rs * (A.v[k] - 4.f * (s0[si] * inv) - (s1[si] * inv) * (rs * (Tg[k] - 4.f * mean[si])));
        }
        o.store(dx + pix * lddx + c);
    }
}

template <typename T>
static int instnorm_spade_bwd_up2_impl(const T* dout, int lddo, const T* z, int ldz, const T* gamma, int ldg, const float* save_mean, const float* save_rstd,
                                       T* dx, int lddx, T* dgamma, int lddg, T* dbeta, int lddb, void* workspace, size_t workspace_bytes,
                                       int N, int Hi, int Wi, int C, const T* xlo, int ldxlo, void* stream) {
    if (!dout || (!z && !xlo) || !gamma || !save_mean || !save_rstd || !dx || !dgamma || !workspace || N < 1 || Hi < 1 || Wi < 1 || C < 1) return MRDIS_EINVAL;
    const long long HW = 4LL * Hi * Wi;
    const size_t pbytes = mrdis_norm_workspace(N, HW, C);
    const size_t need = pbytes + sizeof(float) * 2 * (size_t)N * C;
    if (workspace_bytes < need) return MRDIS_EWORKSPACE;
    const bool v = vec4_ok(dout, lddo, C) && (xlo ? vec4_ok(xlo, ldxlo, C) : vec4_ok(z, ldz, C)) && vec4_ok(gamma, ldg, C) && vec4_ok(dx, lddx, C) &&
                   vec4_ok(dgamma, lddg, C) && (!dbeta || vec4_ok(dbeta, lddb, C));
    // channel chunks of 32: the last one may be narrower; the threads split as (pixel, quad) need NT % quads == 0: quads in {1, 2, 4, 8}
    const int lastq = ((C - 1) % UB_CC + 1) / 4;
    if (!v || N > 65535 || C > 65535 * UB_CC || (lastq & (lastq - 1)) != 0) return MRDIS_EUNSUPPORTED;
    {   // the kernel forms a pixel's element offset inside its image with 24-bit multiplies
        const long long px_img = 4LL * Hi * Wi;
        long long ldmax = lddo > ldg ? lddo : ldg;
        if (lddg > ldmax) ldmax = lddg;
        if (dbeta && lddb > ldmax) ldmax = lddb;
        if (!xlo && ldz > ldmax) ldmax = ldz;
        if (px_img >= (1 << 24) || ldmax >= (1 << 24) || px_img * ldmax >= 0xffffffffLL) return MRDIS_EUNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = mrdis_cdiv(Wi, UB_T), tiles_y = mrdis_cdiv(Hi, UB_T);
    {
        // one pass over the full-resolution tensors (see spade_bwd_up2_kernel ONEPASS): needs x itself and room for one partial pair per (image, tile, channel);
        // bf16 maps: A = U^T dzh goes through an fp32 buffer (rounding it to bf16 before the subtraction of the mean terms would cost the result's leading bits)
        constexpr bool f32 = std::is_same<T, float>::value;
        const int NW = mrdis_opt(MRDIS_OPT_MODE) == 2002 ? 4 : 8;         // waves per workgroup (debug_mode 2002: the 256-thread form, for A/B)
        const size_t p1bytes = sizeof(float) * 2 * (size_t)N * NW * tiles_x * tiles_y * C + 64;
        const size_t abytes = f32 ? 0 : sizeof(float) * (size_t)N * Hi * Wi * C;
        if (xlo && workspace_bytes >= p1bytes + sizeof(float) * 2 * (size_t)N * C + abytes && mrdis_opt(MRDIS_OPT_MODE) != 2001) {
            float* part1 = reinterpret_cast<float*>(workspace);
            float* t0 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + p1bytes);
            float* t1 = t0 + (size_t)N * C;
            float* abuf = f32 ? nullptr : t1 + (size_t)N * C;
            if (NW == 8)       // (reading a stored z instead of interpolating it from xlo was measured level to slower: the kernel is not bound by the interpolation)
                MRDIS_LAUNCH((spade_bwd_up2_kernel<T, true, 512, true>), dim3(tiles_x * tiles_y, N, mrdis_cdiv(C, UB_CC)), dim3(512), 0, s, dout, lddo, z, ldz, gamma, ldg,
                                   save_mean, save_rstd, nullptr, nullptr, dx, lddx, dgamma, lddg, dbeta, lddb, Hi, Wi, C, tiles_x, xlo, ldxlo, part1, abuf,
                                   (mrdis_opt(MRDIS_OPT_MODE) >= 2100 && mrdis_opt(MRDIS_OPT_MODE) < 2200) ? (int)mrdis_opt(MRDIS_OPT_MODE) - 2100 : 0);
            else
                MRDIS_LAUNCH((spade_bwd_up2_kernel<T, true, 256, true>), dim3(tiles_x * tiles_y, N, mrdis_cdiv(C, UB_CC)), dim3(256), 0, s, dout, lddo, z, ldz, gamma, ldg,
                                   save_mean, save_rstd, nullptr, nullptr, dx, lddx, dgamma, lddg, dbeta, lddb, Hi, Wi, C, tiles_x, xlo, ldxlo, part1, abuf);
            MRDIS_CHECK_LAUNCH();
            const int chunks = NW * tiles_x * tiles_y;
            MRDIS_LAUNCH((stat_final_kernel<1>), dim3(mrdis_cdiv(N * C, 64)), dim3(64, stat_final_lanes(chunks)), 0, s, part1, chunks, C, N, HW, 0.f, 0.f, t0, t1, nullptr, nullptr);
            MRDIS_CHECK_LAUNCH();
            MRDIS_LAUNCH((spade_bwd_up2_final_kernel<T>), dim3(ew_blocks((long long)N * Hi * Wi * (C / 4))), dim3(256), 0, s, dx, lddx, xlo, ldxlo, save_mean, save_rstd, t0, t1, N, Hi, Wi, C, abuf);
            MRDIS_CHECK_LAUNCH();
            return MRDIS_OK;
        }
    }
    float* part = reinterpret_cast<float*>(workspace);
    float* s0 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pbytes);
    float* s1 = s0 + (size_t)N * C;
    int rc = launch_stats<2, T>(dout, lddo, z, ldz, gamma, ldg, save_mean, save_rstd, 1, N, HW, C, part, s, 0, xlo, ldxlo, Wi);
    if (rc) return rc;
    const StatPlan sp = stat_plan(N, HW);
    MRDIS_LAUNCH((stat_final_kernel<1>), dim3(mrdis_cdiv(N * C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, N, HW, 0.f, 0.f, s0, s1, nullptr, nullptr);
    MRDIS_CHECK_LAUNCH();
    if (xlo)
        MRDIS_LAUNCH((spade_bwd_up2_kernel<T, false, 512, true>), dim3(tiles_x * tiles_y, N, mrdis_cdiv(C, UB_CC)), dim3(512), 0, s, dout, lddo, z, ldz, gamma, ldg, save_mean, save_rstd, s0, s1,
                           dx, lddx, dgamma, lddg, dbeta, lddb, Hi, Wi, C, tiles_x, xlo, ldxlo);
    else
        MRDIS_LAUNCH((spade_bwd_up2_kernel<T, false, 512, false>), dim3(tiles_x * tiles_y, N, mrdis_cdiv(C, UB_CC)), dim3(512), 0, s, dout, lddo, z, ldz, gamma, ldg, save_mean, save_rstd, s0, s1,
                           dx, lddx, dgamma, lddg, dbeta, lddb, Hi, Wi, C, tiles_x, xlo, ldxlo);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
extern "C" size_t mrdis_instnorm_spade_bwd_workspace(int N, long long HW, int C) {
    return mrdis_norm_workspace(N, HW, C) + sizeof(float) * 2 * (size_t)N * C;
}
extern "C" size_t mrdis_instnorm_spade_bwd_up2_workspace(int N, int Hi, int Wi, int C, int dtype) {
    const size_t two_pass = mrdis_instnorm_spade_bwd_workspace(N, 4LL * Hi * Wi, C);
    const size_t one_pass = sizeof(float) * 2 * (size_t)N * 8 * mrdis_cdiv(Wi, UB_T) * mrdis_cdiv(Hi, UB_T) * C + 64 + sizeof(float) * 2 * (size_t)N * C +
                            (dtype == MRDIS_DT_F32 ? 0 : sizeof(float) * (size_t)N * Hi * Wi * C);
    return one_pass > two_pass ? one_pass : two_pass;
}

// ------------------------------------------------------------------ LeakyReLU backward
template <int V, typename T>
__global__ void lrelu_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy, T* __restrict__ dx, int lddx,
                                 long long P, int C, float slope) {
    const int Q = C / V;
    EW_LOOP(P * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        Vec<V> d, a, o; d.load(dy + r * lddy + c); a.load(y + r * ldy + c);
#pragma unroll
        for (int k = 0; k < V; ++k) o.v[k] = a.v[k] > 0.f ? d.v[k] : slope * d.v[k];
        o.store(dx + r * lddx + c);
    }
}
template <typename T>
static int lrelu_bwd_impl(const T* dy, int lddy, const T* y, int ldy, T* dx, int lddx,
                               long long P, int C, float slope, void* stream) {
    if (!dy || !y || !dx || P < 1 || C < 1) return MRDIS_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (vec4_ok(dy, lddy, C) && vec4_ok(y, ldy, C) && vec4_ok(dx, lddx, C))
        MRDIS_LAUNCH((lrelu_bwd_kernel<4, T>), dim3(ew_blocks(P * C / 4)), dim3(256), 0, s, dy, lddy, y, ldy, dx, lddx, P, C, slope);
    else
        MRDIS_LAUNCH((lrelu_bwd_kernel<1, T>), dim3(ew_blocks(P * C)), dim3(256), 0, s, dy, lddy, y, ldy, dx, lddx, P, C, slope);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ bilinear resize
// Source-index rule of ATen's upsample_bilinear2d (area_pixel_compute_source_index), fp32.
struct BilAxis { int i0, i1; float l0, l1; };
__device__ __forceinline__ BilAxis bil_axis(int o, float scale, int align, int isz) {
    float src;
    if (align) src = scale * (float)o;
    else { src = scale * ((float)o + 0.5f) - 0.5f; if (src < 0.f) src = 0.f; }
    BilAxis a;
    a.i0 = (int)src; if (a.i0 > isz - 1) a.i0 = isz - 1;
    a.i1 = a.i0 + ((a.i0 < isz - 1) ? 1 : 0);
    a.l1 = src - (float)a.i0; a.l0 = 1.f - a.l1;
    return a;
}
static inline float bil_scale(int isz, int osz, int align) {
    if (align) return osz > 1 ? (float)(isz - 1) / (float)(osz - 1) : 0.f;
    return (float)isz / (float)osz;
}

// grid (Ho, N): one output row per workgroup, so the row's vertical stencil is wave-uniform and a thread only
// splits a 32-bit in-row index (the flat-index version spent its time in 64-bit divisions: 1.6 TB/s).
template <int V, typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int Hi, int Wi, int Ho, int Wo,
                                    int C, int align, float sh, float sw) {
    const int Q = C / V;
    const int ho = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x), n = blockIdx.y;   // XCD-aware: neighbouring output rows read the same two input rows -> same L2
    const BilAxis ah = bil_axis(ho, sh, align, Hi);
    const T* r0 = x + ((long long)n * Hi + ah.i0) * Wi * ldx;
    const T* r1 = x + ((long long)n * Hi + ah.i1) * Wi * ldx;
    T* yo = y + ((long long)n * Ho + ho) * Wo * ldy;
    const int items = Wo * Q;
    // two items per trip: eight 16-byte loads in flight per thread (the tail item is clamped to the last one and not stored)
    for (int j = threadIdx.x; j < items; j += 2 * blockDim.x) {
        const int j1 = j + blockDim.x;
        const bool has1 = j1 < items;
        const int jb = has1 ? j1 : j;
        const int wo0 = j / Q, q0 = j - wo0 * Q, wo1 = jb / Q, q1 = jb - wo1 * Q;
        const BilAxis a0 = bil_axis(wo0, sw, align, Wi), a1 = bil_axis(wo1, sw, align, Wi);
        Vec<V> p00, p01, p10, p11, s00, s01, s10, s11, o;
        p00.load(r0 + (long long)a0.i0 * ldx + q0 * V); p01.load(r0 + (long long)a0.i1 * ldx + q0 * V);
        p10.load(r1 + (long long)a0.i0 * ldx + q0 * V); p11.load(r1 + (long long)a0.i1 * ldx + q0 * V);
        s00.load(r0 + (long long)a1.i0 * ldx + q1 * V); s01.load(r0 + (long long)a1.i1 * ldx + q1 * V);
        s10.load(r1 + (long long)a1.i0 * ldx + q1 * V); s11.load(r1 + (long long)a1.i1 * ldx + q1 * V);
#pragma unroll
        for (int k = 0; k < V; ++k)
            o.v[k] = ah.l0 * (a0.l0 * p00.v[k] + a0.l1 * p01.v[k]) + ah.l1 * (a0.l0 * p10.v[k] + a0.l1 * p11.v[k]);
        o.store(yo + (long long)wo0 * ldy + q0 * V);
        if (has1) {
#pragma unroll
            for (int k = 0; k < V; ++k)
                o.v[k] = ah.l0 * (a1.l0 * s00.v[k] + a1.l1 * s01.v[k]) + ah.l1 * (a1.l0 * s10.v[k] + a1.l1 * s11.v[k]);
            o.store(yo + (long long)wo1 * ldy + q1 * V);
        }
    }
}

// gather form of the adjoint: input pixel i collects every output o whose stencil touches it.
__device__ __forceinline__ void bil_range(int i, float scale, int align, int osz, int* lo, int* hi) {
    if (scale <= 0.f) { *lo = 0; *hi = osz - 1; return; }
    float a, b;
    if (align) { a = ((float)i - 1.f) / scale; b = ((float)i + 1.f) / scale; }
    else { a = ((float)i - 0.5f) / scale - 0.5f; b = ((float)i + 1.5f) / scale - 0.5f; }
    int l = (int)floorf(a) - 1, h = (int)ceilf(b) + 1;
    if (l < 0) l = 0; if (h > osz - 1) h = osz - 1;
    *lo = l; *hi = h;
}
// grid (Hi, N): one input row per workgroup (its range of contributing output rows is wave-uniform).  The row
// weights are evaluated once per workgroup and the column weights once per thread (not once per visited output
// pixel): for the x2 up-sampling of the decoders the 7x7 candidate window holds only 3x3 non-zero weights and the
// old loop spent its time re-deriving them.
#define BIL_MAXR 8
template <int V, typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int Hi, int Wi, int Ho, int Wo,
                                    int C, int align, float sh, float sw) {
    const int Q = C / V;
    const int hi = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x), n = blockIdx.y;   // XCD-aware: neighbouring input rows gather from the same output rows
    int hlo, hhi;
    bil_range(hi, sh, align, Ho, &hlo, &hhi);
    const T* base = dy + (long long)n * Ho * Wo * lddy;
    T* dxo = dx + ((long long)n * Hi + hi) * Wi * lddx;
    const int items = Wi * Q;
    const bool small_h = hhi - hlo < BIL_MAXR;
    float whv[BIL_MAXR];
#pragma unroll
    for (int k = 0; k < BIL_MAXR; ++k) {
        whv[k] = 0.f;
        const int ho = hlo + k;
        if (small_h && ho <= hhi) {
            const BilAxis ah = bil_axis(ho, sh, align, Hi);
            whv[k] = (ah.i0 == hi ? ah.l0 : 0.f) + (ah.i1 == hi ? ah.l1 : 0.f);
        }
    }
    for (int j = threadIdx.x; j < items; j += blockDim.x) {
        const int wi = j / Q, q = j - wi * Q;
        int wlo, whi;
        bil_range(wi, sw, align, Wo, &wlo, &whi);
        Vec<V> acc;
#pragma unroll
        for (int k = 0; k < V; ++k) acc.v[k] = 0.f;
        if (small_h && whi - wlo < BIL_MAXR) {
            float wwv[BIL_MAXR];
#pragma unroll
            for (int k = 0; k < BIL_MAXR; ++k) {
                wwv[k] = 0.f;
                const int wo = wlo + k;
                if (wo <= whi) {
                    const BilAxis aw = bil_axis(wo, sw, align, Wi);
                    wwv[k] = (aw.i0 == wi ? aw.l0 : 0.f) + (aw.i1 == wi ? aw.l1 : 0.f);
                }
            }
            // same visiting order (ho ascending, wo ascending) and the same products as the generic loop
#pragma unroll
            for (int a = 0; a < BIL_MAXR; ++a) {
                const float wh = whv[a];
                if (wh == 0.f) continue;
                const T* row = base + (long long)(hlo + a) * Wo * lddy + q * V;
#pragma unroll
                for (int b = 0; b < BIL_MAXR; ++b) {
                    const float ww = wwv[b];
                    if (ww == 0.f) continue;
                    Vec<V> d; d.load(row + (long long)(wlo + b) * lddy);
#pragma unroll
                    for (int k = 0; k < V; ++k) acc.v[k] += wh * ww * d.v[k];
                }
            }
        } else {
            for (int ho = hlo; ho <= hhi; ++ho) {
                const BilAxis ah = bil_axis(ho, sh, align, Hi);
                const float wh = (ah.i0 == hi ? ah.l0 : 0.f) + (ah.i1 == hi ? ah.l1 : 0.f);
                if (wh == 0.f) continue;
                for (int wo = wlo; wo <= whi; ++wo) {
                    const BilAxis aw = bil_axis(wo, sw, align, Wi);
                    const float ww = (aw.i0 == wi ? aw.l0 : 0.f) + (aw.i1 == wi ? aw.l1 : 0.f);
                    if (ww == 0.f) continue;
                    Vec<V> d; d.load(base + ((long long)ho * Wo + wo) * lddy + q * V);
#pragma unroll
                    for (int k = 0; k < V; ++k) acc.v[k] += wh * ww * d.v[k];
                }
            }
        }
        acc.store(dxo + (long long)wi * lddx + q * V);
    }
}

// Branch-free form for the common scales (support of an input pixel <= NA x NB output pixels: 3 x 3 when down-sampling,
// 5 x 5 for the x2 up-sampling of the decoders).  The window above is conservative (7 x 7 for x2) and skips its zero-weight
// candidates with `continue` -- every skipped or taken load is then a branch, and hipcc waits for all outstanding loads at
// each join, so the ~12 useful loads of a thread ran one after the other.  Here the first contributing row / column is
// found by evaluating weights only (no memory), then NA x NB loads are issued unconditionally (coordinates clamped,
// weights zero outside the support): same products, same summation order, all loads in flight.
template <int V, int NA, int NB, typename T>
__global__ void bilinear_bwd_tight_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int Hi, int Wi,
                                          int Ho, int Wo, int C, int align, float sh, float sw) {
    const int Q = C / V;
    const int hi = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x), n = blockIdx.y;
    int hlo, hhi;
    bil_range(hi, sh, align, Ho, &hlo, &hhi);
    auto wgt_h = [&](int ho) { const BilAxis a = bil_axis(ho, sh, align, Hi); return (a.i0 == hi ? a.l0 : 0.f) + (a.i1 == hi ? a.l1 : 0.f); };
    int hs = hlo;
    while (hs < hhi && wgt_h(hs) == 0.f) ++hs;                       // block-uniform
    float whv[NA]; int hrow[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int ho = hs + a;
        hrow[a] = ho <= Ho - 1 ? ho : Ho - 1;
        whv[a] = ho <= hhi ? wgt_h(ho) : 0.f;
    }
    const T* base = dy + (long long)n * Ho * Wo * lddy;
    T* dxo = dx + ((long long)n * Hi + hi) * Wi * lddx;
    const int items = Wi * Q;
    for (int j = threadIdx.x; j < items; j += blockDim.x) {
        const int wi = j / Q, q = j - wi * Q;
        int wlo, whi;
        bil_range(wi, sw, align, Wo, &wlo, &whi);
        auto wgt_w = [&](int wo) { const BilAxis a = bil_axis(wo, sw, align, Wi); return (a.i0 == wi ? a.l0 : 0.f) + (a.i1 == wi ? a.l1 : 0.f); };
        int ws_ = wlo;
#pragma unroll
        for (int k = 0; k < BIL_MAXR; ++k) if (ws_ < whi && wgt_w(ws_) == 0.f) ++ws_;
        float wwv[NB]; int wcol[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int wo = ws_ + b;
            wcol[b] = wo <= Wo - 1 ? wo : Wo - 1;
            wwv[b] = wo <= whi ? wgt_w(wo) : 0.f;
        }
        Vec<V> d[NA][NB];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) d[a][b].load(base + ((long long)hrow[a] * Wo + wcol[b]) * lddy + q * V);
        Vec<V> acc;
#pragma unroll
        for (int k = 0; k < V; ++k) acc.v[k] = 0.f;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                if (whv[a] != 0.f && wwv[b] != 0.f) {                 // value select, no memory access inside
#pragma unroll
                    for (int k = 0; k < V; ++k) acc.v[k] += whv[a] * wwv[b] * d[a][b].v[k];
                }
        acc.store(dxo + (long long)wi * lddx + q * V);
    }
}

// ---- exact x2 up-sampling, align_corners = False (nn.Upsample(scale_factor=(2,2)) of the SPADE decoder, model.py:2551): the
// source index rule gives out[2i] = 0.25 x[max(i-1,0)] + 0.75 x[i], out[2i+1] = 0.75 x[i] + 0.25 x[min(i+1,H-1)] on both axes
// (same fp32 products and association as the generic kernel).  A thread owns one INPUT pixel (4 channels) and writes its 2x2
// outputs from the 3x3 neighbourhood: 9 loads per 4 outputs instead of 16, no per-output index arithmetic.
// STATS: the kernel also leaves the per-(image, channel) sums of the values it STORES (sum, sum of squares; one partial per input row in
// the `part` layout of stat_final_kernel, chunk = input row): the consumer of the up-sampled map is a SPADE block's InstanceNorm
// (model.py:2440), whose statistics pass would otherwise read the 4x tensor back from HBM.  Needs blockDim.x % (C / 4) == 0 (a thread
// keeps its channel group); block reduction in LDS in a fixed order.
template <typename T, bool STATS = false>
__global__ void bilinear_up2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int Hi, int Wi, int C, float* __restrict__ part = nullptr,
                                        int oblk = 0x7fffffff, long long oblk_stride = 0) {      // output image n lives in block n / oblk (blocks oblk_stride elements apart)
    __shared__ float red_[STATS ? 256 * 9 : 1];
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    const int Q = C / 4;
    const int i = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x), n = blockIdx.y;
    const int im = i > 0 ? i - 1 : 0, ip = i < Hi - 1 ? i + 1 : Hi - 1;
    const T* rm = x + ((long long)n * Hi + im) * Wi * ldx;
    const T* rc = x + ((long long)n * Hi + i) * Wi * ldx;
    const T* rp = x + ((long long)n * Hi + ip) * Wi * ldx;
    T* y0 = y + (long long)(n / oblk) * oblk_stride + ((long long)(n % oblk) * 2 * Hi + 2 * i) * (2 * Wi) * ldy;
    T* y1 = y0 + (long long)(2 * Wi) * ldy;
    // row weights of the two output rows: (top, centre, bottom); at the borders the clamped neighbour IS the centre row
    const float a_t = i > 0 ? 0.25f : 0.f, a_c0 = i > 0 ? 0.75f : 1.f;              // out[2i]   = a_t x[im] + a_c0 x[i]   (src clamped to 0 at i = 0)
    const float a_b = 0.25f, a_c1 = 0.75f;                                          // out[2i+1] = a_c1 x[i] + a_b x[ip]   (ip = i at the last row)
    const int items = Wi * Q;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int j = it / Q, q = it - j * Q;
        const int jm = j > 0 ? j - 1 : 0, jp = j < Wi - 1 ? j + 1 : Wi - 1;
        const float b_l = j > 0 ? 0.25f : 0.f, b_c0 = j > 0 ? 0.75f : 1.f, b_r = 0.25f, b_c1 = 0.75f;
        Vec<4> tl, tc, tr, cl_, cc, cr, bl, bc, br, o;
        tl.load(rm + (long long)jm * ldx + 4 * q); tc.load(rm + (long long)j * ldx + 4 * q); tr.load(rm + (long long)jp * ldx + 4 * q);
        cl_.load(rc + (long long)jm * ldx + 4 * q); cc.load(rc + (long long)j * ldx + 4 * q); cr.load(rc + (long long)jp * ldx + 4 * q);
        bl.load(rp + (long long)jm * ldx + 4 * q); bc.load(rp + (long long)j * ldx + 4 * q); br.load(rp + (long long)jp * ldx + 4 * q);
        // generic form: l0h * (l0w * p00 + l1w * p01) + l1h * (l0w * p10 + l1w * p11), (p0x = upper row, px0 = left column)
        auto tally = [&](const Vec<4>& v_) {
            if (STATS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float r_ = (float)(T)v_.v[k]; s0[k] += r_; s1[k] += r_ * r_; }      // the value as stored
            }
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) o.v[k] = a_t * (b_l * tl.v[k] + b_c0 * tc.v[k]) + a_c0 * (b_l * cl_.v[k] + b_c0 * cc.v[k]);
        o.store(y0 + (long long)(2 * j) * ldy + 4 * q); tally(o);
#pragma unroll
        for (int k = 0; k < 4; ++k) o.v[k] = a_t * (b_c1 * tc.v[k] + b_r * tr.v[k]) + a_c0 * (b_c1 * cc.v[k] + b_r * cr.v[k]);
        o.store(y0 + (long long)(2 * j + 1) * ldy + 4 * q); tally(o);
#pragma unroll
        for (int k = 0; k < 4; ++k) o.v[k] = a_c1 * (b_l * cl_.v[k] + b_c0 * cc.v[k]) + a_b * (b_l * bl.v[k] + b_c0 * bc.v[k]);
        o.store(y1 + (long long)(2 * j) * ldy + 4 * q); tally(o);
#pragma unroll
        for (int k = 0; k < 4; ++k) o.v[k] = a_c1 * (b_c1 * cc.v[k] + b_r * cr.v[k]) + a_b * (b_c1 * bc.v[k] + b_r * br.v[k]);
        o.store(y1 + (long long)(2 * j + 1) * ldy + 4 * q); tally(o);
    }
    if (STATS) {
        // every thread's items share one channel group q = threadIdx.x % Q; the blockDim.x / Q threads of a group are added in thread order
        const int tid = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) { red_[tid * 9 + k] = s0[k]; red_[tid * 9 + 4 + k] = s1[k]; }
        __syncthreads();
        if (tid < Q) {
            float t_[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int u = tid; u < (int)blockDim.x; u += Q)
#pragma unroll
                for (int k = 0; k < 8; ++k) t_[k] += red_[u * 9 + k];
            float* dst = part + ((long long)(n * Hi + i) * 2) * C;          // group = image n, chunk = input row i
#pragma unroll
            for (int k = 0; k < 4; ++k) { dst[4 * tid + k] = t_[k]; dst[C + 4 * tid + k] = t_[4 + k]; }
        }
    }
}
// adjoint: dx[i][j] = sum over the 4 x 4 output block rows 2i-1 .. 2i+2, columns 2j-1 .. 2j+2 with the separable weights
// (0.25, 0.75, 0.75, 0.25); at a border the row / column that falls outside is replaced by the clamped one (whose forward weight
// went to the border pixel).  16 loads per input pixel instead of the 25 of the generic tight kernel.
template <typename T>
__global__ void bilinear_up2_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int Hi, int Wi, int C) {
    const int Q = C / 4;
    const int i = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x), n = blockIdx.y;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    int rr[4] = {2 * i - 1, 2 * i, 2 * i + 1, 2 * i + 2};
    float wr[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    if (i == 0) { rr[0] = 0; }                         // out[0] = 1.0 x[0]: the missing 0.25 comes from row 0 itself
    if (i == Hi - 1) { rr[3] = Ho - 1; }               // out[Ho-1] = 1.0 x[Hi-1]
    const T* base = dy + (long long)n * Ho * Wo * lddy;
    T* dxo = dx + ((long long)n * Hi + i) * Wi * lddx;
    const int items = Wi * Q;
    for (int it = threadIdx.x; it < items; it += blockDim.x) {
        const int j = it / Q, q = it - j * Q;
        int cc[4] = {2 * j - 1, 2 * j, 2 * j + 1, 2 * j + 2};
        if (j == 0) cc[0] = 0;
        if (j == Wi - 1) cc[3] = Wo - 1;
        Vec<4> d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[a][b].load(base + ((long long)rr[a] * Wo + cc[b]) * lddy + 4 * q);
        Vec<4> acc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                s += wr[a] * ((0.25f * d[a][0].v[k] + 0.75f * d[a][1].v[k]) + (0.75f * d[a][2].v[k] + 0.25f * d[a][3].v[k]));
            acc.v[k] = s;
        }
        acc.store(dxo + (long long)j * lddx + 4 * q);
    }
}

static inline int bil_threads(long long items) { return items >= 256 ? 256 : (items > 64 ? 128 : 64); }

template <typename T>
static int bilinear_fwd_impl(const T* x, int ldx, T* y, int ldy, int N, int Hi, int Wi,
                                  int Ho, int Wo, int C, int align_corners, void* stream) {
    if (!x || !y || N < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || C < 1 || ldx < C || ldy < C) return MRDIS_EINVAL;
    if (N > 65535) return MRDIS_EUNSUPPORTED;
    const float sh = bil_scale(Hi, Ho, align_corners), sw = bil_scale(Wi, Wo, align_corners);
    hipStream_t s = (hipStream_t)stream;
    if (!align_corners && Ho == 2 * Hi && Wo == 2 * Wi && vec4_ok(x, ldx, C) && vec4_ok(y, ldy, C) && !mrdis_opt(MRDIS_OPT_BILGEN)) {
        MRDIS_LAUNCH((bilinear_up2_fwd_kernel<T>), dim3(Hi, N), dim3(bil_threads((long long)Wi * (C / 4))), 0, s, x, ldx, y, ldy, Hi, Wi, C);
        MRDIS_CHECK_LAUNCH();
        return MRDIS_OK;
    }
    if (vec4_ok(x, ldx, C) && vec4_ok(y, ldy, C))
        MRDIS_LAUNCH((bilinear_fwd_kernel<4, T>), dim3(Ho, N), dim3(bil_threads((long long)Wo * (C / 4))), 0, s, x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, align_corners, sh, sw);
    else
        MRDIS_LAUNCH((bilinear_fwd_kernel<1, T>), dim3(Ho, N), dim3(bil_threads((long long)Wo * C)), 0, s, x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, align_corners, sh, sw);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
template <typename T>
static int bilinear_bwd_impl(const T* dy, int lddy, T* dx, int lddx, int N, int Hi, int Wi,
                                  int Ho, int Wo, int C, int align_corners, void* stream) {
    if (!dy || !dx || N < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || C < 1 || lddy < C || lddx < C) return MRDIS_EINVAL;
    if (N > 65535) return MRDIS_EUNSUPPORTED;
    const float sh = bil_scale(Hi, Ho, align_corners), sw = bil_scale(Wi, Wo, align_corners);
    hipStream_t s = (hipStream_t)stream;
    // support of an input pixel along one axis: < 2 / scale + 1 output pixels (scale = input step per output pixel)
    const float smin = sh < sw ? sh : sw;
    const bool big = (long long)N * Hi * Wi * C >= 6000000LL;          // measured: 108 vs 129 us at 16 M elements, a wash below 6 M
    if (!align_corners && Ho == 2 * Hi && Wo == 2 * Wi && vec4_ok(dy, lddy, C) && vec4_ok(dx, lddx, C) && !mrdis_opt(MRDIS_OPT_BILGEN)) {
        MRDIS_LAUNCH((bilinear_up2_bwd_kernel<T>), dim3(Hi, N), dim3(bil_threads((long long)Wi * (C / 4))), 0, s, dy, lddy, dx, lddx, Hi, Wi, C);
        MRDIS_CHECK_LAUNCH();
        return MRDIS_OK;
    }
    const bool tight3 = big && sh >= 1.f && sw >= 1.f, tight5 = big && smin > 0.4975f && !mrdis_opt(MRDIS_OPT_BILGEN);
    if (vec4_ok(dy, lddy, C) && vec4_ok(dx, lddx, C) && tight3 && !mrdis_opt(MRDIS_OPT_BILGEN))
        MRDIS_LAUNCH((bilinear_bwd_tight_kernel<4, 3, 3, T>), dim3(Hi, N), dim3(bil_threads((long long)Wi * (C / 4))), 0, s, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, align_corners, sh, sw);
    else if (vec4_ok(dy, lddy, C) && vec4_ok(dx, lddx, C) && tight5)
        MRDIS_LAUNCH((bilinear_bwd_tight_kernel<4, 5, 5, T>), dim3(Hi, N), dim3(bil_threads((long long)Wi * (C / 4))), 0, s, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, align_corners, sh, sw);
    else if (vec4_ok(dy, lddy, C) && vec4_ok(dx, lddx, C))
        MRDIS_LAUNCH((bilinear_bwd_kernel<4, T>), dim3(Hi, N), dim3(bil_threads((long long)Wi * (C / 4))), 0, s, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, align_corners, sh, sw);
    else
        MRDIS_LAUNCH((bilinear_bwd_kernel<1, T>), dim3(Hi, N), dim3(bil_threads((long long)Wi * C)), 0, s, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, align_corners, sh, sw);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// nn.Upsample(scale_factor=(2,2), bilinear, align_corners=False) + the instance statistics of its result in one pass (see the STATS form
// of bilinear_up2_fwd_kernel): y as mrdis_bilinear_fwd writes it, save_mean / save_rstd as mrdis_instnorm_stats would compute them from y
// (same fp64 combine of fp32 partial sums; the partial sums are taken in another order).
extern "C" size_t mrdis_bilinear_up2_stats_workspace(int N, int Hi, int C) { return sizeof(float) * 2 * (size_t)N * Hi * C + 64; }
template <typename T>
static int bilinear_up2_stats_impl(const T* x, int ldx, T* y, int ldy, int N, int Hi, int Wi, int C, int out_block, long long out_block_stride,
                                   float* save_mean, float* save_rstd, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !y || !save_mean || !save_rstd || !workspace || N < 1 || Hi < 1 || Wi < 1 || C < 1 || ldx < C || ldy < C) return MRDIS_EINVAL;
    if (out_block <= 0 || out_block >= N) { out_block = 0x7fffffff; out_block_stride = 0; }       // dense output
    else if (N % out_block != 0 || out_block_stride < 4LL * out_block * Hi * Wi * ldy || out_block_stride % 4 != 0) return MRDIS_EINVAL;
    if (workspace_bytes < mrdis_bilinear_up2_stats_workspace(N, Hi, C)) return MRDIS_EWORKSPACE;
    const int threads = bil_threads((long long)Wi * (C / 4));
    if (N > 65535 || C % 4 != 0 || !vec4_ok(x, ldx, C) || !vec4_ok(y, ldy, C) || threads % (C / 4) != 0) return MRDIS_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    MRDIS_LAUNCH((bilinear_up2_fwd_kernel<T, true>), dim3(Hi, N), dim3(threads), 0, s, x, ldx, y, ldy, Hi, Wi, C, part, out_block, out_block_stride);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH((stat_final_kernel<0>), dim3(mrdis_cdiv(N * C, 64)), dim3(64, stat_final_lanes(Hi)), 0, s, part, Hi, C, N, 4LL * Hi * Wi, eps, 0.f,
                       save_mean, save_rstd, nullptr, nullptr);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// 1 where mrdis_bilinear_up2_stats_fwd takes the geometry (dense, 16-byte aligned views assumed), 0 where it returns MRDIS_EUNSUPPORTED: the block size it
// picks must hold whole rows of C / 4 channel groups, the grid's y extent is N
extern "C" int mrdis_bilinear_up2_stats_applies(int N, int Wi, int C) {
    if (N < 1 || N > 65535 || Wi < 1 || C < 4 || C % 4 != 0) return 0;
    return bil_threads((long long)Wi * (C / 4)) % (C / 4) == 0 ? 1 : 0;
}

// ------------------------------------------------------------------ softmax([scale*mask, s])[1:]

// ---- C ABI: activation views are fp32 or bf16 by `dtype` (include/mrdis.h MRDIS_DT_*); statistics, parameters and their
// gradients are always fp32
typedef const __bf16* cbf; typedef __bf16* bf;
extern "C" int mrdis_bn_train_fwd(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                  long long P, int C, float eps, float momentum, int groups, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype,
        bn_train_fwd_impl((const float*)x, ldx, (float*)y, ldy, gamma, beta, running_mean, running_var, save_mean, save_rstd, workspace, workspace_bytes, P, C, eps, momentum, groups, stream),
        bn_train_fwd_impl((cbf)x, ldx, (bf)y, ldy, gamma, beta, running_mean, running_var, save_mean, save_rstd, workspace, workspace_bytes, P, C, eps, momentum, groups, stream));
}
extern "C" int mrdis_bn_eval_fwd(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, long long P, int C, float eps, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype, bn_eval_fwd_impl((const float*)x, ldx, (float*)y, ldy, gamma, beta, running_mean, running_var, P, C, eps, stream),
                          bn_eval_fwd_impl((cbf)x, ldx, (bf)y, ldy, gamma, beta, running_mean, running_var, P, C, eps, stream));
}
extern "C" int mrdis_bn_train_bwd(const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* save_mean,
                                  const float* save_rstd, void* dx, int lddx, float* dgamma, float* dbeta, float* acc_dgamma, float* acc_dbeta,
                                  void* workspace, size_t workspace_bytes, long long P, int C, int groups, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype,
        bn_train_bwd_impl((const float*)dy, lddy, (const float*)x, ldx, gamma, save_mean, save_rstd, (float*)dx, lddx, dgamma, dbeta, acc_dgamma, acc_dbeta, workspace, workspace_bytes, P, C, groups, stream),
        bn_train_bwd_impl((cbf)dy, lddy, (cbf)x, ldx, gamma, save_mean, save_rstd, (bf)dx, lddx, dgamma, dbeta, acc_dgamma, acc_dbeta, workspace, workspace_bytes, P, C, groups, stream));
}
extern "C" int mrdis_instnorm_spade_fwd(const void* z, int ldz, const void* gamma, int ldg, const void* beta, int ldb, void* out, int ldo,
                                        float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                        int N, long long HW, int C, float eps, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype,
        instnorm_spade_fwd_impl((const float*)z, ldz, (const float*)gamma, ldg, (const float*)beta, ldb, (float*)out, ldo, save_mean, save_rstd, workspace, workspace_bytes, N, HW, C, eps, stream),
        instnorm_spade_fwd_impl((cbf)z, ldz, (cbf)gamma, ldg, (cbf)beta, ldb, (bf)out, ldo, save_mean, save_rstd, workspace, workspace_bytes, N, HW, C, eps, stream));
}
// Instance statistics only (mean, 1 / sqrt(var + eps) per (sample, channel)): the first half of mrdis_instnorm_spade_fwd, for callers that
// apply the modulation elsewhere (the SPADE epilogue of the Winograd kernel, mrdis_conv2d_fwd_spade).
extern "C" int mrdis_instnorm_stats(const void* z, int ldz, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                    int N, long long HW, int C, float eps, int dtype, void* stream) {
    if (!z || !save_mean || !save_rstd || !workspace || N < 1 || HW < 1 || C < 1) return MRDIS_EINVAL;
    if (workspace_bytes < mrdis_norm_workspace(N, HW, C)) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    int rc;
    if (dtype == MRDIS_DT_BF16) rc = launch_stats<0, __bf16>((const __bf16*)z, ldz, (const __bf16*)nullptr, 0, (const __bf16*)nullptr, 0, nullptr, nullptr, 0, N, HW, C, part, s);
    else rc = launch_stats<0, float>((const float*)z, ldz, (const float*)nullptr, 0, (const float*)nullptr, 0, nullptr, nullptr, 0, N, HW, C, part, s);
    if (rc) return rc;
    const StatPlan sp = stat_plan(N, HW);
    MRDIS_LAUNCH((stat_final_kernel<0>), dim3(mrdis_cdiv(N * C, 64)), dim3(64, stat_final_lanes(sp.chunks)), 0, s, part, sp.chunks, C, N, HW, eps, 0.f,
                       save_mean, save_rstd, nullptr, nullptr);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
extern "C" int mrdis_instnorm_spade_bwd(const void* dout, int lddo, const void* z, int ldz, const void* gamma, int ldg, const float* save_mean,
                                        const float* save_rstd, void* dz, int lddz, void* dgamma, int lddg, void* dbeta, int lddb,
                                        void* workspace, size_t workspace_bytes, int N, long long HW, int C, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype,
        instnorm_spade_bwd_impl((const float*)dout, lddo, (const float*)z, ldz, (const float*)gamma, ldg, save_mean, save_rstd, (float*)dz, lddz, (float*)dgamma, lddg, (float*)dbeta, lddb, workspace, workspace_bytes, N, HW, C, stream),
        instnorm_spade_bwd_impl((cbf)dout, lddo, (cbf)z, ldz, (cbf)gamma, ldg, save_mean, save_rstd, (bf)dz, lddz, (bf)dgamma, lddg, (bf)dbeta, lddb, workspace, workspace_bytes, N, HW, C, stream));
}
extern "C" int mrdis_instnorm_spade_bwd_up2(const void* dout, int lddo, const void* z, int ldz, const void* gamma, int ldg, const float* save_mean,
                                            const float* save_rstd, void* dx, int lddx, void* dgamma, int lddg, void* dbeta, int lddb,
                                            void* workspace, size_t workspace_bytes, int N, int Hi, int Wi, int C, const void* xlo, int ldxlo, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype,
        instnorm_spade_bwd_up2_impl((const float*)dout, lddo, (const float*)z, ldz, (const float*)gamma, ldg, save_mean, save_rstd, (float*)dx, lddx, (float*)dgamma, lddg, (float*)dbeta, lddb, workspace, workspace_bytes, N, Hi, Wi, C, (const float*)xlo, ldxlo, stream),
        instnorm_spade_bwd_up2_impl((cbf)dout, lddo, (cbf)z, ldz, (cbf)gamma, ldg, save_mean, save_rstd, (bf)dx, lddx, (bf)dgamma, lddg, (bf)dbeta, lddb, workspace, workspace_bytes, N, Hi, Wi, C, (cbf)xlo, ldxlo, stream));
}
extern "C" int mrdis_lrelu_bwd(const void* dy, int lddy, const void* y, int ldy, void* dx, int lddx, long long P, int C, float slope, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype, lrelu_bwd_impl((const float*)dy, lddy, (const float*)y, ldy, (float*)dx, lddx, P, C, slope, stream),
                          lrelu_bwd_impl((cbf)dy, lddy, (cbf)y, ldy, (bf)dx, lddx, P, C, slope, stream));
}
extern "C" int mrdis_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype, bilinear_fwd_impl((const float*)x, ldx, (float*)y, ldy, N, Hi, Wi, Ho, Wo, C, align_corners, stream),
                          bilinear_fwd_impl((cbf)x, ldx, (bf)y, ldy, N, Hi, Wi, Ho, Wo, C, align_corners, stream));
}
extern "C" int mrdis_bilinear_up2_stats_fwd(const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int C, int out_block, long long out_block_stride,
                                            float* save_mean, float* save_rstd, float eps, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype,
        bilinear_up2_stats_impl((const float*)x, ldx, (float*)y, ldy, N, Hi, Wi, C, out_block, out_block_stride, save_mean, save_rstd, eps, workspace, workspace_bytes, stream),
        bilinear_up2_stats_impl((cbf)x, ldx, (bf)y, ldy, N, Hi, Wi, C, out_block, out_block_stride, save_mean, save_rstd, eps, workspace, workspace_bytes, stream));
}
extern "C" int mrdis_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int dtype, void* stream) {
    return MRDIS_BY_DTYPE(dtype, bilinear_bwd_impl((const float*)dy, lddy, (float*)dx, lddx, N, Hi, Wi, Ho, Wo, C, align_corners, stream),
                          bilinear_bwd_impl((cbf)dy, lddy, (bf)dx, lddx, N, Hi, Wi, Ho, Wo, C, align_corners, stream));
}

// NHWC view cast between the two storage types (src -> dst, P rows), optionally changing the channel count: the first
// min(C_src, C_dst) channels are copied, the rest of a wider destination is ZERO.  The boundary between bf16 activations and
// the fp32-only tensors (4-channel anatomy maps, 7-channel inputs / reconstructions): a narrow fp32 view is padded to a 16-channel
// bf16 one so that the bf16 MFMA kernels take the layer, and the padded result is sliced back.
template <int V, typename TS, typename TD>
__global__ void cast_view_kernel(const TS* __restrict__ src, int lds_, int Cs, TD* __restrict__ dst, int ldd, int Cd, long long P) {
    const int Q = Cd / V;
    EW_LOOP(P * Q) {
        const long long r = idx / Q; const int c = (int)(idx - r * Q) * V;
        Vec<V> a;
        if (c + V <= Cs) a.load(src + r * lds_ + c);
        else {
#pragma unroll
            for (int k = 0; k < V; ++k) a.v[k] = (c + k < Cs) ? ld1(src + r * lds_ + c + k) : 0.f;
        }
        a.store(dst + r * ldd + c);
    }
}
template <typename TS, typename TD>
static int cast_view_impl(const TS* src, int lds_, int Cs, TD* dst, int ldd, int Cd, long long P, void* stream) {
    if (!src || !dst || P < 1 || Cs < 1 || Cd < 1 || lds_ < Cs || ldd < Cd) return MRDIS_EINVAL;
    if (vec4_ok(dst, ldd, Cd) && (Cs >= Cd ? vec4_ok(src, lds_, Cd) : (lds_ % 4 == 0 && Cs % 4 == 0 && (((uintptr_t)src & (4 * sizeof(TS) - 1)) == 0))))
        MRDIS_LAUNCH((cast_view_kernel<4, TS, TD>), dim3(ew_blocks(P * Cd / 4)), dim3(256), 0, (hipStream_t)stream, src, lds_, Cs, dst, ldd, Cd, P);
    else
        MRDIS_LAUNCH((cast_view_kernel<1, TS, TD>), dim3(ew_blocks(P * Cd)), dim3(256), 0, (hipStream_t)stream, src, lds_, Cs, dst, ldd, Cd, P);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
extern "C" int mrdis_cast_view(const void* src, int ld_src, int src_dtype, int C_src, void* dst, int ld_dst, int dst_dtype, int C_dst,
                               long long P, void* stream) {
    const bool sb = src_dtype == MRDIS_DT_BF16, db = dst_dtype == MRDIS_DT_BF16;
    if (sb && !db) return cast_view_impl((cbf)src, ld_src, C_src, (float*)dst, ld_dst, C_dst, P, stream);
    if (!sb && db) return cast_view_impl((const float*)src, ld_src, C_src, (bf)dst, ld_dst, C_dst, P, stream);
    if (sb && db) return cast_view_impl((cbf)src, ld_src, C_src, (bf)dst, ld_dst, C_dst, P, stream);
    return cast_view_impl((const float*)src, ld_src, C_src, (float*)dst, ld_dst, C_dst, P, stream);
}

#define SM_MAXC 8
__global__ void softmax_md_fwd_kernel(const float* __restrict__ s, int lds_, const float* __restrict__ mask, float* __restrict__ out, int ldo,
                                      long long P, int C, float scale) {
    EW_LOOP(P) {
        float l[SM_MAXC];
        const float l0 = mask ? scale * mask[idx] : -INFINITY;
        float mx = l0;
        for (int c = 0; c < C; ++c) { l[c] = s[idx * lds_ + c]; mx = fmaxf(mx, l[c]); }
        float den = mask ? expf(l0 - mx) : 0.f;
        for (int c = 0; c < C; ++c) { l[c] = expf(l[c] - mx); den += l[c]; }
        const float inv = 1.f / den;
        for (int c = 0; c < C; ++c) out[idx * ldo + c] = l[c] * inv;
    }
}
__global__ void softmax_md_bwd_kernel(const float* __restrict__ dout, int lddo, const float* __restrict__ out, int ldo, float* __restrict__ ds, int ldds,
                                      long long P, int C) {
    EW_LOOP(P) {
        float o[SM_MAXC], d[SM_MAXC];
        float dot = 0.f;
        for (int c = 0; c < C; ++c) { o[c] = out[idx * ldo + c]; d[c] = dout[idx * lddo + c]; dot += o[c] * d[c]; }
        for (int c = 0; c < C; ++c) ds[idx * ldds + c] = o[c] * (d[c] - dot);
    }
}
extern "C" int mrdis_softmax_mask_drop_fwd(const float* s, int lds_, const float* mask_img, float* out, int ldo,
                                           long long P, int C, float mask_scale, void* stream) {
    if (!s || !out || P < 1 || C < 1 || C > SM_MAXC) return C > SM_MAXC ? MRDIS_EUNSUPPORTED : MRDIS_EINVAL;
    MRDIS_LAUNCH(softmax_md_fwd_kernel, dim3(ew_blocks(P)), dim3(256), 0, (hipStream_t)stream, s, lds_, mask_img, out, ldo, P, C, mask_scale);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
extern "C" int mrdis_softmax_mask_drop_bwd(const float* dout, int lddo, const float* out, int ldo,
                                           float* ds, int ldds, long long P, int C, void* stream) {
    if (!dout || !out || !ds || P < 1 || C < 1 || C > SM_MAXC) return C > SM_MAXC ? MRDIS_EUNSUPPORTED : MRDIS_EINVAL;
    MRDIS_LAUNCH(softmax_md_bwd_kernel, dim3(ew_blocks(P)), dim3(256), 0, (hipStream_t)stream, dout, lddo, out, ldo, ds, ldds, P, C);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ batch assembly (ZeroDoseDataset.__getitem__, util.py:471-566)
// Volumes stay resident in HBM as [D][H][W] planes (a BraTS fold is ~28 GB of fp32: 10 % of one MI355X), so a batch
// is a device-side gather instead of h5 reads + numpy concatenation + a 110 MB host-to-device copy per step:
//   inputs[b][h][w][m*(2*block+1) + k] = vol(b, m)[slice_b - block + k][h][w]   (zeros if the contrast is missing
//                                         for that subject or was dropped, util.py:523-525 / :540-542)
//   mask[b][m] = contrast present and not dropped;   mask_img[b][h][w] = (inputs[b][0][h][w] == 0)  (:563)
// vol_ptrs: (B, M) device pointers (0 = missing); slice_idx already clamped to [block, D-1-block] (:476-484).
__global__ void slice_gather_kernel(const unsigned long long* __restrict__ vol_ptrs, const int* __restrict__ slice_idx,
                                    const int* __restrict__ drop, float* __restrict__ inputs, int ld, float* __restrict__ mask,
                                    float* __restrict__ mask_img, int B, int M, int H, int W, int D, int block) {
    const int c7 = 2 * block + 1;
    const long long HW = (long long)H * W;
    const int b = blockIdx.y;
    const int s0 = slice_idx[b] - block;
    const int dm = drop[b];
    if (blockIdx.x == 0 && blockIdx.z == 0 && (int)threadIdx.x < M) {
        const int mm = threadIdx.x;
        mask[b * M + mm] = (vol_ptrs[b * M + mm] != 0ull && mm != dm) ? 1.f : 0.f;
    }
    // grid (x, B, M): a wave reads 64 consecutive pixels of one plane (coalesced) and writes their 2*block+1 channels
    const int m = blockIdx.z;
    const float* vol = reinterpret_cast<const float*>(vol_ptrs[b * M + m]);
    const bool live = vol != nullptr && m != dm;
    for (long long px = blockIdx.x * (long long)blockDim.x + threadIdx.x; px < HW; px += (long long)gridDim.x * blockDim.x) {
        float* dst = inputs + ((long long)b * HW + px) * ld + m * c7;
        float first = 0.f;
        for (int k = 0; k < c7; ++k) {
            const int sk = s0 + k;
            const float v = (live && sk >= 0 && sk < D) ? vol[(long long)sk * HW + px] : 0.f;
            dst[k] = v;
            if (k == 0) first = v;
        }
        if (m == 0) mask_img[(long long)b * HW + px] = (first == 0.f) ? 1.f : 0.f;
    }
}
// all channels of a pixel in one thread (C = M*(2*block+1) <= 32, C % 4 == 0, 16-byte rows): every plane read is a
// coalesced 256-byte wave access and the pixel leaves as C/4 16-byte stores, consecutive lanes writing consecutive
// pixels -- the per-contrast version above writes 28-byte pieces at a 112-byte stride (0.6 TB/s).
template <int CMAX>
__global__ void slice_gather_px_kernel(const unsigned long long* __restrict__ vol_ptrs, const int* __restrict__ slice_idx,
                                       const int* __restrict__ drop, float* __restrict__ inputs, int ld, float* __restrict__ mask,
                                       float* __restrict__ mask_img, int B, int M, int H, int W, int D, int block) {
    const int c7 = 2 * block + 1, C = M * c7;
    const long long HW = (long long)H * W;
    const int b = blockIdx.y;
    const int s0 = slice_idx[b] - block;
    const int dm = drop[b];
    if (blockIdx.x == 0 && (int)threadIdx.x < M) {
        const int mm = threadIdx.x;
        mask[b * M + mm] = (vol_ptrs[b * M + mm] != 0ull && mm != dm) ? 1.f : 0.f;
    }
    for (long long px = blockIdx.x * (long long)blockDim.x + threadIdx.x; px < HW; px += (long long)gridDim.x * blockDim.x) {
        float v[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            v[c] = 0.f;
            if (c < C) {
                const int m = c / c7, k = c - m * c7;
                const float* vol = reinterpret_cast<const float*>(vol_ptrs[b * M + m]);       // wave-uniform
                const int sk = s0 + k;
                if (vol != nullptr && m != dm && sk >= 0 && sk < D) v[c] = vol[(long long)sk * HW + px];
            }
        }
        float* dst = inputs + ((long long)b * HW + px) * ld;
#pragma unroll
        for (int c = 0; c < CMAX; c += 4)
            if (c < C) *reinterpret_cast<float4*>(dst + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
        mask_img[(long long)b * HW + px] = (v[0] == 0.f) ? 1.f : 0.f;
    }
}
extern "C" int mrdis_slice_gather(const void* vol_ptrs, const int* slice_idx, const int* drop, float* inputs, int ld_in,
                                  float* mask, float* mask_img, int B, int M, int H, int W, int D, int block, void* stream) {
    if (!vol_ptrs || !slice_idx || !drop || !inputs || !mask || !mask_img || B < 1 || M < 1 || M > 64 || H < 1 || W < 1 || D < 1 ||
        block < 0 || 2 * block + 1 > D || ld_in < M * (2 * block + 1)) return MRDIS_EINVAL;
    if (B > 65535) return MRDIS_EUNSUPPORTED;
    const long long items = (long long)H * W;
    int gx = (int)((items + 255) / 256); if (gx > 1024) gx = 1024;
    const int C = M * (2 * block + 1);
    if (C % 4 == 0 && C <= 32 && ld_in % 4 == 0 && (((uintptr_t)inputs & 15) == 0)) {
        MRDIS_LAUNCH(slice_gather_px_kernel<32>, dim3(gx, B), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const unsigned long long*>(vol_ptrs), slice_idx, drop, inputs, ld_in, mask, mask_img, B, M, H, W, D, block);
        MRDIS_CHECK_LAUNCH();
        return MRDIS_OK;
    }
    MRDIS_LAUNCH(slice_gather_kernel, dim3(gx, B, M), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(vol_ptrs), slice_idx, drop, inputs, ld_in, mask, mask_img, B, M, H, W, D, block);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ reconstruction metrics (evaluate(), util.py:935-978)
// One image = channel 0 of one sample.  Both images are shifted by their own minimum, data range R =
// max of the shifted target; MSE, PSNR = 10 log10(R^2/MSE), SSIM = mean over the 7x7-valid region of
// the structural-similarity map (uniform window, sample covariance NP/(NP-1), K1 = 0.01, K2 = 0.03).
#define MET_WIN 7
#define MET_MAXW 1024
__global__ void metrics_minmax_kernel(const float* __restrict__ t, int ldt, const float* __restrict__ x, int ldx, long long HW,
                                      float* __restrict__ mm) {
    __shared__ float red[3][4];
    const long long base = (long long)blockIdx.x * HW;
    float tmin = INFINITY, tmax = -INFINITY, xmin = INFINITY;
    for (long long e = threadIdx.x; e < HW; e += blockDim.x) {
        const float a = t[(base + e) * ldt], b = x[(base + e) * ldx];
        tmin = fminf(tmin, a); tmax = fmaxf(tmax, a); xmin = fminf(xmin, b);
    }
    for (int o = 32; o > 0; o >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, o)); tmax = fmaxf(tmax, __shfl_xor(tmax, o)); xmin = fminf(xmin, __shfl_xor(xmin, o));
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = tmin; red[1][threadIdx.x >> 6] = tmax; red[2][threadIdx.x >> 6] = xmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mm[blockIdx.x * 3 + 0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
        mm[blockIdx.x * 3 + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        mm[blockIdx.x * 3 + 2] = fminf(fminf(red[2][0], red[2][1]), fminf(red[2][2], red[2][3]));
    }
}
// grid (H, n_img): block y = one image row.  Row sums of squared error for every row; SSIM-map row
// sums for the rows whose 7x7 window fits.  part[(img*H + row)*2 + {0,1}] (fp64).
__global__ void metrics_rows_kernel(const float* __restrict__ t, int ldt, const float* __restrict__ x, int ldx, int H, int W,
                                    const float* __restrict__ mm, double* __restrict__ part) {
    extern __shared__ float rows[];                      // [MET_WIN][2][W]
    __shared__ double red[2][4];
    const int img = blockIdx.y, y = blockIdx.x;
    const float tmin = mm[img * 3 + 0], xmin = mm[img * 3 + 2];
    const double R = (double)mm[img * 3 + 1] - (double)tmin;
    const long long base = (long long)img * H * W;
    const int pad = MET_WIN / 2;
    const bool has_win = (y >= pad) && (y < H - pad) && (W >= MET_WIN);
    const int y0 = has_win ? y - pad : y, nrow = has_win ? MET_WIN : 1;
    for (int e = threadIdx.x; e < nrow * W; e += blockDim.x) {
        const int r = e / W, c = e - r * W;
        const long long p = base + (long long)(y0 + r) * W + c;
        rows[(r * 2 + 0) * W + c] = t[p * ldt] - tmin;
        rows[(r * 2 + 1) * W + c] = x[p * ldx] - xmin;
    }
    __syncthreads();
    const int rc = has_win ? pad : 0;                    // this row inside the staged window
    double se = 0.0, ss = 0.0;
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        const double d = (double)rows[(rc * 2 + 0) * W + c] - (double)rows[(rc * 2 + 1) * W + c];
        se += d * d;
        if (has_win && c >= pad && c < W - pad) {
            double sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
            for (int r = 0; r < MET_WIN; ++r)
                for (int k = -pad; k <= pad; ++k) {
                    const double a = rows[(r * 2 + 0) * W + c + k], b = rows[(r * 2 + 1) * W + c + k];
                    sa += a; sb += b; saa += a * a; sbb += b * b; sab += a * b;
                }
            const double NP = MET_WIN * MET_WIN, cn = NP / (NP - 1.0);
            const double ua = sa / NP, ub = sb / NP;
            const double va = cn * (saa / NP - ua * ua), vb = cn * (sbb / NP - ub * ub), vab = cn * (sab / NP - ua * ub);
            const double C1 = (0.01 * R) * (0.01 * R), C2 = (0.03 * R) * (0.03 * R);
            ss += ((2 * ua * ub + C1) * (2 * vab + C2)) / ((ua * ua + ub * ub + C1) * (va + vb + C2));
        }
    }
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o); ss += __shfl_xor(ss, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = se; red[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((long long)img * H + y) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[((long long)img * H + y) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ void metrics_final_kernel(const double* __restrict__ part, const float* __restrict__ mm, int n_img, int H, int W,
                                     float* __restrict__ out) {
    const int img = blockIdx.x * blockDim.x + threadIdx.x;
    if (img >= n_img) return;
    double se = 0.0, ss = 0.0;
    for (int y = 0; y < H; ++y) { se += part[((long long)img * H + y) * 2]; ss += part[((long long)img * H + y) * 2 + 1]; }
    const double R = (double)mm[img * 3 + 1] - (double)mm[img * 3 + 0];
    const double mse = se / ((double)H * W);
    out[img * 3 + 0] = (float)mse;
    out[img * 3 + 1] = (float)(10.0 * log10(R * R / mse));
    out[img * 3 + 2] = (H >= MET_WIN && W >= MET_WIN) ? (float)(ss / ((double)(H - MET_WIN + 1) * (W - MET_WIN + 1))) : NAN;
}
extern "C" size_t mrdis_recon_metrics_workspace(int n_img, int H) {
    return sizeof(double) * 2 * (size_t)n_img * H + sizeof(float) * 4 * (size_t)n_img + 64;
}
extern "C" int mrdis_recon_metrics(const float* target, int ldt, const float* pred, int ldp, float* out,
                                   void* workspace, size_t workspace_bytes, int n_img, int H, int W, void* stream) {
    if (!target || !pred || !out || !workspace || n_img < 1 || H < 1 || W < 1 || ldt < 1 || ldp < 1) return MRDIS_EINVAL;
    if (W > MET_MAXW || n_img > 65535) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes < mrdis_recon_metrics_workspace(n_img, H)) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(workspace);
    float* mm = reinterpret_cast<float*>(part + 2 * (size_t)n_img * H);
    MRDIS_LAUNCH(metrics_minmax_kernel, dim3(n_img), dim3(256), 0, s, target, ldt, pred, ldp, (long long)H * W, mm);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(metrics_rows_kernel, dim3(H, n_img), dim3(256), sizeof(float) * MET_WIN * 2 * W, s, target, ldt, pred, ldp, H, W, mm, part);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(metrics_final_kernel, dim3(mrdis_cdiv(n_img, 64)), dim3(64), 0, s, part, mm, n_img, H, W, out);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ reconstruction error (per-sample mean)
__global__ void recon_partial_kernel(const float* __restrict__ gt, int ldgt, const float* __restrict__ x, int ldx, long long HW, int C, int p,
                                     int epb, float* __restrict__ part) {
    __shared__ float red[4];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const long long tot = HW * C;
    const long long e0 = (long long)chunk * epb;
    long long e1 = e0 + epb; if (e1 > tot) e1 = tot;
    float s = 0.f;
    for (long long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const long long r = e / C; const int c = (int)(e - r * C);
        const long long row = (long long)n * HW + r;
        const float d = gt[row * ldgt + c] - x[row * ldx + c];
        s += (p == 1) ? fabsf(d) : d * d;
    }
    s = mrdis_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)n * gridDim.x + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void recon_final_kernel(const float* __restrict__ part, int chunks, int N, double inv, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += (double)part[(long long)n * chunks + k];
    out[n] = (float)(s * inv);
}
static int recon_chunks(int N, long long tot, int* epb) {
    long long maxc = 4096 / (N > 0 ? N : 1); if (maxc < 1) maxc = 1;
    long long chunks = (tot + 4095) / 4096; if (chunks > maxc) chunks = maxc; if (chunks < 1) chunks = 1;
    *epb = (int)((tot + chunks - 1) / chunks);
    return (int)chunks;
}
extern "C" size_t mrdis_recon_err_workspace(int N, long long HW, int C) {
    int epb; const int ch = recon_chunks(N, HW * C, &epb);
    return sizeof(float) * (size_t)N * ch + 64;
}
extern "C" int mrdis_recon_err_fwd(const float* gt, int ldgt, const float* x, int ldx, float* out,
                                   void* workspace, size_t workspace_bytes,
                                   int N, long long HW, int C, int p, void* stream) {
    if (!gt || !x || !out || !workspace || N < 1 || HW < 1 || C < 1 || (p != 1 && p != 2)) return MRDIS_EINVAL;
    if (workspace_bytes < mrdis_recon_err_workspace(N, HW, C)) return MRDIS_EWORKSPACE;
    int epb; const int ch = recon_chunks(N, HW * C, &epb);
    hipStream_t s = (hipStream_t)stream;
    float* part = reinterpret_cast<float*>(workspace);
    MRDIS_LAUNCH(recon_partial_kernel, dim3(ch, N), dim3(256), 0, s, gt, ldgt, x, ldx, HW, C, p, epb, part);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(recon_final_kernel, dim3(mrdis_cdiv(N, 64)), dim3(64), 0, s, part, ch, N, 1.0 / ((double)HW * C), out);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
__global__ void recon_bwd_kernel(const float* __restrict__ gt, int ldgt, const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                 float* __restrict__ dx, int lddx, int N, long long HW, int C, int p, float inv) {
    EW_LOOP((long long)N * HW * C) {
        const long long r = idx / C; const int c = (int)(idx - r * C);
        const int n = (int)(r / HW);
        const float d = x[r * ldx + c] - gt[r * ldgt + c];
        const float g = (p == 1) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d;
        dx[r * lddx + c] = w[n] * inv * g;
    }
}
extern "C" int mrdis_recon_err_bwd(const float* gt, int ldgt, const float* x, int ldx, const float* w,
                                   float* dx, int lddx, int N, long long HW, int C, int p, void* stream) {
    if (!gt || !x || !w || !dx || N < 1 || HW < 1 || C < 1 || (p != 1 && p != 2)) return MRDIS_EINVAL;
    MRDIS_LAUNCH(recon_bwd_kernel, dim3(ew_blocks((long long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, gt, ldgt, x, ldx, w, dx, lddx,
                       N, HW, C, p, (float)(1.0 / ((double)HW * C)));
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ max-pool k x k, stride k (floor)
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int32_t* __restrict__ arg, int N, int H, int W, int C,
                                   int k, int Ho, int Wo) {
    EW_LOOP((long long)N * Ho * Wo * C) {
        const int c = (int)(idx % C); long long r = idx / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho); const int n = (int)(r / Ho);
        float best = -INFINITY; int bi = (ho * k) * W + wo * k;
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j) {
                const int pix = (ho * k + i) * W + (wo * k + j);
                const float v = x[((long long)n * H * W + pix) * ldx + c];
                if (v > best || v != v) { best = v; bi = pix; }       // first maximum wins (ATen scan order)
            }
        y[idx] = best; arg[idx] = bi;
    }
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg, float* __restrict__ dx, int lddx, int N, int H, int W,
                                   int C, int k, int Ho, int Wo) {
    // one thread per element of dx: dy of its window where it was the maximum, 0 elsewhere and in the rows / columns the floor leaves out.  Every element
    // is written by exactly one thread: no memset in front (a hipMemsetAsync node recorded into a HIP graph did not run again on later replays, round 6)
    EW_LOOP((long long)N * H * W * C) {
        const int c = (int)(idx % C); long long r = idx / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int n = (int)(r / H);
        const int ho = h / k, wo = w / k;
        float v = 0.f;
        if (ho < Ho && wo < Wo) {
            const long long o = (((long long)n * Ho + ho) * Wo + wo) * C + c;
            if (arg[o] == h * W + w) v = dy[o];
        }
        dx[((long long)n * H * W + (long long)h * W + w) * lddx + c] = v;
    }
}
extern "C" int mrdis_maxpool_fwd(const float* x, int ldx, float* y, int32_t* argmax, int N, int H, int W, int C,
                                 int k, void* stream) {
    if (!x || !y || !argmax || N < 1 || C < 1 || k < 1 || H < k || W < k) return MRDIS_EINVAL;
    const int Ho = H / k, Wo = W / k;
    MRDIS_LAUNCH(maxpool_fwd_kernel, dim3(ew_blocks((long long)N * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, argmax, N, H, W, C, k, Ho, Wo);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
extern "C" int mrdis_maxpool_bwd(const float* dy, const int32_t* argmax, float* dx, int lddx, int N, int H, int W,
                                 int C, int k, void* stream) {
    if (!dy || !argmax || !dx || N < 1 || C < 1 || k < 1 || H < k || W < k) return MRDIS_EINVAL;
    const int Ho = H / k, Wo = W / k;
    hipStream_t s = (hipStream_t)stream;
    if (lddx < C) return MRDIS_EINVAL;
    MRDIS_LAUNCH(maxpool_bwd_kernel, dim3(ew_blocks((long long)N * H * W * C)), dim3(256), 0, s, dy, argmax, dx, lddx, N, H, W, C, k, Ho, Wo);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ optimizer arena
__global__ void sumsq_finite_kernel(const float* __restrict__ g, long long n, float* __restrict__ part) {
    __shared__ double red[4]; __shared__ float bad[4];
    double s = 0.0; float nb = 0.f;
    EW_LOOP(n) { const float v = g[idx]; if (!(fabsf(v) <= 3.402823466e38f)) nb += 1.f; else s += (double)v * (double)v; }
    s = mrdis_wave_sum_d(s); nb = mrdis_wave_sum(nb);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; bad[threadIdx.x >> 6] = nb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
        part[2 * blockIdx.x + 1] = (bad[0] + bad[1]) + (bad[2] + bad[3]);
    }
}
__global__ void sumsq_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double s = 0.0, b = 0.0;
    for (int k = 0; k < nblk; ++k) { s += (double)part[2 * k]; b += (double)part[2 * k + 1]; }
    out[0] += (float)s; out[1] += (float)b;
}
#define SUMSQ_BLOCKS 1024
extern "C" size_t mrdis_sumsq_workspace(void) { return sizeof(float) * 2 * SUMSQ_BLOCKS; }
extern "C" int mrdis_sumsq_finite(const float* g, long long n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!g || !out || !workspace || n < 1) return MRDIS_EINVAL;
    if (workspace_bytes < mrdis_sumsq_workspace()) return MRDIS_EWORKSPACE;
    int nb = ew_blocks(n); if (nb > SUMSQ_BLOCKS) nb = SUMSQ_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    MRDIS_LAUNCH(sumsq_finite_kernel, dim3(nb), dim3(256), 0, s, g, n, reinterpret_cast<float*>(workspace));
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(sumsq_final_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const float*>(workspace), nb, out);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// torch.optim.Adam(amsgrad=True, weight_decay=wd) single-tensor rule, fp32:
//   g += wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; vmax = max(vmax, v)
//   p -= (lr / (1-b1^t)) * m / (sqrt(vmax)/sqrt(1-b2^t) + eps)
// Segment gates: torch's Adam skips a parameter whose gradient is None (a decoder whose modality is absent from the whole
// batch); in the arena that is an index range, gated by a device flag (0 = no gradient arrived: leave p, m, v, vmax alone).
#define ADAM_MAX_GATES 32
struct AdamGates { int n; long long lo[ADAM_MAX_GATES], hi[ADAM_MAX_GATES]; int flag[ADAM_MAX_GATES]; };

// step_state[0] = optimizer steps applied so far, [1] = steps skipped because a gradient was non-finite.
// gate_steps (float[3 * n_flags] or NULL): [k] = steps applied to gate group k -- torch.optim.Adam keeps `step` PER PARAMETER and does
// not advance it while the parameter's gradient is None, so a decoder whose modality was absent for some batches is bias-corrected
// with ITS count, not the arena's; [n_flags + 2k], [n_flags + 2k + 1] = that group's (1 - b1^t, sqrt(1 - b2^t)), written here so that
// the element kernel loads two floats instead of evaluating powf per element.
__global__ void adam_advance_kernel(float* __restrict__ step_state, const float* __restrict__ norm_finite,
                                    float* __restrict__ gate_steps, const float* __restrict__ gate_flags, int n_flags, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool skipped = norm_finite && norm_finite[1] > 0.f;
    if (skipped) step_state[1] += 1.f; else step_state[0] += 1.f;
    if (gate_steps && gate_flags) {
        for (int k = 0; k < n_flags; ++k) {
            float t = gate_steps[k];
            if (!skipped && gate_flags[k] != 0.f) { t += 1.f; gate_steps[k] = t; }
            gate_steps[n_flags + 2 * k] = 1.f - powf(b1, t);
            gate_steps[n_flags + 2 * k + 1] = sqrtf(1.f - powf(b2, t));
        }
    }
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float* __restrict__ vmax,
                            long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float sbc2,
                            const float* __restrict__ norm_finite, float max_norm, float gscale,
                            const float* __restrict__ step_state, const float* __restrict__ gate_flags, AdamGates gates,
                            const float* __restrict__ gate_steps, int n_flags) {
    float coef = gscale;
    if (norm_finite) {
        if (norm_finite[1] > 0.f) return;                       // non-finite gradient: skip the step (main_missing.py:273-278)
        if (max_norm > 0.f) {
            const float tn = sqrtf(norm_finite[0]) * gscale;
            const float cc = max_norm / (tn + 1e-6f);            // clip_grad_norm_ (main_missing.py:272)
            coef = gscale * (cc < 1.f ? cc : 1.f);
        }
    }
    if (step_state) {                                           // bias corrections from the device-side step counter
        const float t = step_state[0];
        bc1 = 1.f - powf(b1, t); sbc2 = sqrtf(1.f - powf(b2, t));
    }
    const float step0 = lr / bc1, sbc20 = sbc2;
    EW_LOOP(n) {
        float step = step0;
        sbc2 = sbc20;
        if (gate_flags) {
            bool off = false;
            int gk = -1;
            for (int k = 0; k < gates.n; ++k)
                if (idx >= gates.lo[k] && idx < gates.hi[k]) { gk = gates.flag[k]; off |= gate_flags[gk] == 0.f; }
            if (off) continue;
            if (gk >= 0 && gate_steps) {                            // this range's own step count (torch: per-parameter `step`)
                step = lr / gate_steps[n_flags + 2 * gk];
                sbc2 = gate_steps[n_flags + 2 * gk + 1];
            }
        }
        const float pv = p[idx];
        const float gv = g[idx] * coef + wd * pv;
        const float mv = b1 * m[idx] + (1.f - b1) * gv;
        const float vv = b2 * v[idx] + (1.f - b2) * gv * gv;
        const float vm = fmaxf(vmax[idx], vv);
        m[idx] = mv; v[idx] = vv; vmax[idx] = vm;
        p[idx] = pv - step * mv / (sqrtf(vm) / sbc2 + eps);
    }
}
extern "C" int mrdis_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax,
                                       long long n, float lr, float beta1, float beta2, float eps,
                                       float weight_decay, int step_count, float* step_state, const float* norm_finite,
                                       float max_norm, float grad_scale, const long long* gate_ranges, const int* gate_flag_index,
                                       int n_gates, const float* gate_flags, float* gate_steps, int n_flags, void* stream) {
    if (!p || !g || !m || !v || !vmax || n < 1 || (!step_state && step_count < 1)) return MRDIS_EINVAL;
    if (n_gates < 0 || n_gates > ADAM_MAX_GATES || (n_gates > 0 && (!gate_ranges || !gate_flag_index || !gate_flags))) return MRDIS_EINVAL;
    if (gate_steps && (n_gates == 0 || !step_state || n_flags < 1 || n_flags > ADAM_MAX_GATES)) return MRDIS_EINVAL;
    for (int k = 0; k < n_gates; ++k)
        if (gate_flag_index[k] < 0 || (gate_steps && gate_flag_index[k] >= n_flags)) return MRDIS_EINVAL;
    AdamGates gates{};
    gates.n = n_gates;
    for (int k = 0; k < n_gates; ++k) { gates.lo[k] = gate_ranges[2 * k]; gates.hi[k] = gate_ranges[2 * k + 1]; gates.flag[k] = gate_flag_index[k]; }
    float bc1 = 1.f, sbc2 = 1.f;
    if (step_state) {
        MRDIS_LAUNCH(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_state, norm_finite,
                           gate_steps, n_gates > 0 ? gate_flags : (const float*)nullptr, n_flags, beta1, beta2);
        MRDIS_CHECK_LAUNCH();
    } else {
        bc1 = 1.f - powf(beta1, (float)step_count);
        sbc2 = sqrtf(1.f - powf(beta2, (float)step_count));
    }
    MRDIS_LAUNCH(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sbc2, norm_finite, max_norm, grad_scale, (const float*)step_state,
                       n_gates > 0 ? gate_flags : (const float*)nullptr, gates, (const float*)gate_steps, n_flags);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ small host -> device transfers without the copy engine
// src may be PINNED HOST memory (mapped into the device's address space): the kernel reads it across the link.  A stream-ordered
// hipMemcpyAsync of a few KB between two kernels costs the GPU ~0.5 ms of idle time (the runtime waits for the preceding kernel on
// the host side before it programs the copy: 10 such copies per training step = 5.5 ms of a 191 ms step); a kernel is just the next
// packet in the queue.
__global__ void copy_bytes_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long long nwords) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nwords; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
extern "C" int mrdis_copy_bytes(const void* src, void* dst, long long nbytes, void* stream) {
    if (!src || !dst || nbytes < 1 || (nbytes & 3) != 0 || (((uintptr_t)src | (uintptr_t)dst) & 3) != 0) return MRDIS_EINVAL;
    const long long nw = nbytes / 4;
    long long nb = (nw + 255) / 256; if (nb > 1024) nb = 1024;
    MRDIS_LAUNCH(copy_bytes_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const unsigned*>(src),
                       reinterpret_cast<unsigned*>(dst), nw);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// Streaming-store probe: n_vec4 float4 written with non-temporal 16-byte stores, nothing read -- what the memory system takes when a kernel ONLY writes
// (bench.py times it over the north-star conv's own output buffers: that layer is 89 % stores, so this, not the 8 TB/s of the data sheet, is its ceiling).
__global__ __launch_bounds__(256) void stream_fill_kernel(f32x4* __restrict__ dst, long long n_vec4, float value) {
    // a workgroup owns one contiguous run of the buffer (XCD-aware: neighbouring runs on the same XCD), a wave-instruction writes 1 KB of it
    const f32x4 v = {value, value, value, value};
    const long long per = (n_vec4 + gridDim.x - 1) / gridDim.x;
    const long long b = mrdis_xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const long long lo = b * per, hi = lo + per < n_vec4 ? lo + per : n_vec4;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) __builtin_nontemporal_store(v, dst + i);
}
extern "C" int mrdis_stream_fill(float* dst, long long n_floats, float value, void* stream) {
    if (!dst || n_floats < 4 || (n_floats & 3) != 0 || (((uintptr_t)dst) & 15) != 0) return MRDIS_EINVAL;
    const long long nv = n_floats / 4;
    long long nb = (nv + 255) / 256; if (nb > 2048) nb = 2048;
    MRDIS_LAUNCH(stream_fill_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4*>(dst), nv, value);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ------------------------------------------------------------------ misc
// ---------------------------------------------------------------------------------------------- process-wide switches
namespace {
struct OptDef { const char* name; const char* env; int is_flag; long long dflt; };
const OptDef OPT_DEFS[MRDIS_OPT_COUNT] = {
    {"wino", "MRDIS_WINO", 0, 1},            // 0 direct kernels only | 1 measured policy | 2 Winograd wherever it applies
    {"nt_mb", "MRDIS_NT_MB", 0, 128},        // outputs of at least this many MB leave the Winograd kernel with non-temporal stores
    {"wino_pipe", "MRDIS_WINO_PIPE", 0, 1},  // 1: the software-pipelined Winograd kernel (mrdis_wino2.hip) where it applies | 0: the phase-by-phase one
    {"wino_u", "MRDIS_WINO_U", 0, 1},        // 1: the pipelined kernel reads a pre-transformed filter image when the caller passes one | 0: always transforms the taps itself
    {"wino4", "MRDIS_WINO4", 0, 1},          // 1: F(4x4, 3x3) (mrdis_wino4.hip) for the filters mrdis_wino_u_format() names, where the grid fills the chip | 0: never | 2: wherever the kernel applies
    {"wino4r", "MRDIS_WINO4R", 0, 1},        // <= 32 couts: 1: the register-fed F(4x4, 3x3) form (mrdis_wino4r.hip) for <= 64 reduction channels and inputs beyond the Infinity Cache | 0: the shared-transform form | 2 / 3: always its 64-tile / channel-split form
    {"bconv4", "MRDIS_BCONV4", 0, 1},        // bf16 3x3 stride-1 layers: 1: the LDS-DMA kernel (mrdis_bf16q.hip) where the launch fills the chip | 0: bconv3_kernel (mrdis_bf16p.hip) | 2: wherever it applies
    {"split6", "MRDIS_SPLIT6", 0, 1},        // thin fp32 3x3 stride-1 layers on the bf16 matrix pipe: both fp32 operands as three bf16 terms, the six products of order <= 2 summed in fp32 (2^-23 relative per product).  1: the 4 -> C kernel for <= 32 couts, the C -> 4 kernel, sp6.out forward + weight gradient (32 -> 16) | 0: fp32 MFMA | 2 .. 7: one 2-D kernel at a time, see run_c4conv | 8 / 9: only the 3-D 16 -> 16 forward + data gradient / weight gradient kernel (mrdis_conv3d_s6.hip; both also under 1) | 10: only the tap-table kernel for launches that bring a filter image (mrdis_s6conv.hip; also under 1)
    {"debug_no16", "MRDIS_DEBUG_NO16", 1, 0}, {"debug_nothin", "MRDIS_DEBUG_NOTHIN", 1, 0}, {"debug_noc4", "MRDIS_DEBUG_NOC4", 1, 0},
    {"debug_nodma", "MRDIS_DEBUG_NODMA", 1, 0}, {"debug_no16_3d", "MRDIS_DEBUG_NO16_3D", 1, 0},
    {"debug_bilgen", "MRDIS_DEBUG_BILGEN", 1, 0}, {"debug_now16", "MRDIS_DEBUG_NOW16", 1, 0},
    {"debug_nopack", "MRDIS_DEBUG_NOPACK", 1, 0},   // 1: the four parity classes of a stride-2 data gradient as four launches
    {"debug_mode", "MRDIS_DEBUG_MODE", 0, -1}, {"debug_bn", "MRDIS_DEBUG_BN", 0, -1}, {"debug_kc", "MRDIS_DEBUG_KC", 0, -1},
    {"debug_bm", "MRDIS_DEBUG_BM", 0, -1}, {"debug_c4_tw", "MRDIS_DEBUG_C4_TW", 0, -1}, {"debug_wgsplit", "MRDIS_DEBUG_WGSPLIT", 0, -1},
    {"debug_bn3", "MRDIS_DEBUG_BN3", 0, -1}, {"debug_kc3", "MRDIS_DEBUG_KC3", 0, -1},
    {"c4_grid", "MRDIS_C4_GRID", 0, 0},      // persistent grid of the Cin = 4 kernels (run_c4conv): 0 = workgroups per CU from the occupancy query of the launched instantiation | k > 0: k workgroups per CU
    {"debug_c4_blocks", "MRDIS_DEBUG_C4_BLOCKS", 0, -1},      // read-only diagnostic: grid.x of the last run_c4conv launch
};
long long* opt_table() {
    static long long* table = [] {
        static long long v[MRDIS_OPT_COUNT];
        for (int i = 0; i < MRDIS_OPT_COUNT; ++i) {
            const char* e = getenv(OPT_DEFS[i].env);
            v[i] = !e ? OPT_DEFS[i].dflt : (OPT_DEFS[i].is_flag ? 1 : atoll(e));
        }
        return v;
    }();
    return table;
}
int opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < MRDIS_OPT_COUNT; ++i)
        if (!strcmp(name, OPT_DEFS[i].name)) return i;
    return -1;
}
}  // namespace
long long mrdis_opt(int id) { return opt_table()[id]; }
void mrdis_opt_note(int id, long long value) { opt_table()[id] = value; }
extern "C" int mrdis_set_option(const char* name, long long value) {
    const int i = opt_index(name);
    if (i < 0) return MRDIS_EINVAL;
    opt_table()[i] = value;
    return MRDIS_OK;
}
extern "C" long long mrdis_get_option(const char* name) {
    const int i = opt_index(name);
    return i < 0 ? (long long)MRDIS_EINVAL : opt_table()[i];
}

namespace {
const char* const CNT_NAMES[MRDIS_CNT_COUNT] = {"wino", "wino_spade", "wino2", "wino2_spade", "wino4", "wino4_spade", "wino4n", "wino4r",
                                                "wino_wgrad", "wino_wgrad2", "wino4_wgrad", "bconv3", "bconv3_spade", "bconv4", "bconv4_spade",
                                                "split6_c4", "split6_c16", "split6_wgrad16", "split6_co4", "split6_c3d", "split6_w3d", "split6_tap", "all"};
long long g_counts[MRDIS_CNT_COUNT];
}  // namespace
void mrdis_count(int id) { __atomic_fetch_add(&g_counts[id], 1LL, __ATOMIC_RELAXED); }
extern "C" long long mrdis_launch_count(const char* family) {
    if (!family) return MRDIS_EINVAL;
    for (int i = 0; i < MRDIS_CNT_COUNT; ++i)
        if (!strcmp(family, CNT_NAMES[i])) return __atomic_load_n(&g_counts[i], __ATOMIC_RELAXED);
    return MRDIS_EINVAL;
}
extern "C" void mrdis_launch_count_reset(void) {
    for (int i = 0; i < MRDIS_CNT_COUNT; ++i) __atomic_store_n(&g_counts[i], 0LL, __ATOMIC_RELAXED);
}

namespace {
struct LdsNote { const char* expr; size_t bytes; };
LdsNote g_lds[128]; int g_nlds = 0;
std::mutex g_lds_mu;        // forward (main thread) and backward (autograd thread) both launch
}  // namespace
void mrdis_note_lds(const char* kernel_expr, size_t bytes) {       // host, launch path: a pointer compare per known kernel (string literals are unique per call site)
    const int n = __atomic_load_n(&g_nlds, __ATOMIC_ACQUIRE);
    for (int i = 0; i < n; ++i)        // known kernel at a size already seen: no lock (entries are only ever appended, bytes only ever grow)
        if (g_lds[i].expr == kernel_expr && bytes <= g_lds[i].bytes) return;
    std::lock_guard<std::mutex> lk(g_lds_mu);
    for (int i = 0; i < g_nlds; ++i)
        if (g_lds[i].expr == kernel_expr) { if (bytes > g_lds[i].bytes) g_lds[i].bytes = bytes; return; }
    if (g_nlds < 128) { g_lds[g_nlds].expr = kernel_expr; g_lds[g_nlds].bytes = bytes; __atomic_store_n(&g_nlds, g_nlds + 1, __ATOMIC_RELEASE); }
}
// "kernel expression=bytes" lines, at most cap - 1 characters, only whole lines; returns the number of entries written
extern "C" int mrdis_dynamic_lds_table(char* buf, int cap) {
    int pos = 0, written = 0;
    if (!buf || cap < 1) return MRDIS_EINVAL;
    buf[0] = 0;
    std::lock_guard<std::mutex> lk(g_lds_mu);
    for (int i = 0; i < g_nlds; ++i) {
        const int n = snprintf(buf + pos, (size_t)(cap - pos), "%s=%zu\n", g_lds[i].expr, g_lds[i].bytes);
        if (n < 0 || pos + n >= cap) { buf[pos] = 0; break; }      // the entry did not fit: drop its truncated text
        pos += n; ++written;
    }
    return written;
}

extern "C" const char* mrdis_strerror(int code) {
    switch (code) {
        case MRDIS_OK: return "ok";
        case MRDIS_EINVAL: return "invalid argument";
        case MRDIS_EUNSUPPORTED: return "unsupported geometry";
        case MRDIS_EWORKSPACE: return "workspace too small";
        case MRDIS_ELAUNCH: return "kernel launch failed";
        case MRDIS_EALIGN: return "misaligned pointer or leading dimension";
        default: return "unknown error";
    }
}
extern "C" int mrdis_version(void) { return 110; }
