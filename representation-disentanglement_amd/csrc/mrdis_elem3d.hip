// mrdis_elem3d.hip -- the bandwidth-bound operators of the 3-D networks (src/model.py:1856-2060) on NDHWC fp32 rows:
// GroupNorm(8) + ReLU (BasicBlock, VAEBranch.hidden_conv) and nearest x2 upsampling fused with the skip addition
// (UNet3D.forward: `u = up(u) + c`).  A tensor is a (N, P = D*H*W, C) row matrix with row pitch ld.
//
// GroupNorm statistics: per-(sample, channel) partial sums over row chunks (float4 lanes, fp32 inside a chunk), combined
// over chunks and the channels of a group in double in a fixed order -> bit-reproducible, no atomics.
#include "mrdis_common.h"

#define GN_CHUNK_ROWS 2048

// partial[n][chunk][C][2]: MODE 0: (sum x, sum x^2) ; MODE 1: (sum dz, sum dz * xhat) with dz = dy * (y > 0)
template <int MODE>
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ partial, long long P, int C, int G, int relu) {
    __shared__ float red[256][9];
    const int C4 = C >> 2, RL = 256 / C4;
    const int tid = threadIdx.x, q = tid % C4, rl = tid / C4;
    const int chunk = blockIdx.x, n = blockIdx.y, nchunk = gridDim.x;
    const long long r0 = (long long)chunk * GN_CHUNK_ROWS;
    long long r1 = r0 + GN_CHUNK_ROWS; if (r1 > P) r1 = P;
    const float* xb = x + (long long)n * P * ldx + 4 * q;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < RL) {
        if (MODE == 0) {
            for (long long r = r0 + rl; r < r1; r += RL) {
                const float4 v = *reinterpret_cast<const float4*>(xb + r * ldx);
                a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
                b[0] += v.x * v.x; b[1] += v.y * v.y; b[2] += v.z * v.z; b[3] += v.w * v.w;
            }
        } else {
            const int cg = C / G;
            const float* dyb = dy + (long long)n * P * lddy + 4 * q;
            float mu[4], rs[4], ga[4], be[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * q + k, g = c / cg;
                mu[k] = mean[n * G + g]; rs[k] = rstd[n * G + g]; ga[k] = gamma[c]; be[k] = beta[c];
            }
            for (long long r = r0 + rl; r < r1; r += RL) {
                const float4 v = *reinterpret_cast<const float4*>(xb + r * ldx);
                const float4 d = *reinterpret_cast<const float4*>(dyb + r * lddy);
                const float xv[4] = {v.x, v.y, v.z, v.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xh = (xv[k] - mu[k]) * rs[k];
                    const float dz = (relu && !(xh * ga[k] + be[k] > 0.f)) ? 0.f : dv[k];       // the forward's expression: same mask
                    a[k] += dz; b[k] += dz * xh;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[tid][k] = a[k]; red[tid][4 + k] = b[k]; }
    __syncthreads();
    if (tid < C4) {            // fixed-order sum over the row lanes
        float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < RL; ++l)
#pragma unroll
            for (int k = 0; k < 4; ++k) { sa[k] += red[l * C4 + tid][k]; sb[k] += red[l * C4 + tid][4 + k]; }
        float* o = partial + (((long long)n * nchunk + chunk) * C + 4 * tid) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[2 * k] = sa[k]; o[2 * k + 1] = sb[k]; }
    }
}

// chan[n][c][2] = sum over chunks: one wave per (n, c), lane-strided partial sums in double, then a shuffle tree (fixed order)
__global__ __launch_bounds__(64) void gn_chan_kernel(const float* __restrict__ partial, double* __restrict__ chan, int N, int nchunk, int C) {
    const int i = blockIdx.x;
    const int n = i / C, c = i - n * C;
    double a = 0.0, b = 0.0;
    const float* src = partial + ((long long)n * nchunk * C + c) * 2;
    for (int k = threadIdx.x; k < nchunk; k += 64) {
        const float2 v = *reinterpret_cast<const float2*>(src + (long long)k * 2 * C);
        a += v.x; b += v.y;
    }
    a = mrdis_wave_sum_d(a); b = mrdis_wave_sum_d(b);
    if (threadIdx.x == 0) { chan[2 * i] = a; chan[2 * i + 1] = b; }
}

__global__ void gn_stat_kernel(const double* __restrict__ chan, float* __restrict__ mean, float* __restrict__ rstd,
                               int N, int C, int G, long long P, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * G) return;
    const int n = i / G, g = i - n * G, cg = C / G;
    double a = 0.0, b = 0.0;
    for (int c = g * cg; c < (g + 1) * cg; ++c) { a += chan[2 * (n * C + c)]; b += chan[2 * (n * C + c) + 1]; }
    const double m = (double)P * cg;
    const double mu = a / m;
    double var = b / m - mu * mu; if (var < 0.0) var = 0.0;
    mean[i] = (float)mu;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = relu?((x - mean) * rstd * gamma + beta) (the centred form: x * sc + sh cancels badly when |mean| >> std).  grid (row
// blocks, N): a thread keeps its channel quad, so the per-channel constants live in registers and a block streams rows
// with four 16-byte loads in flight per thread.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       long long P, int C, int G, int rpb, int relu) {
    const int C4 = C >> 2, RL = 256 / C4, cg = C / G;
    const int tid = threadIdx.x, q = tid % C4, rl = tid / C4, n = blockIdx.y;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + k, g = c / cg;
        mu[k] = mean[n * G + g]; rs[k] = rstd[n * G + g]; ga[k] = gamma[c]; be[k] = beta[c];
    }
    const long long r0 = (long long)blockIdx.x * rpb;
    long long r1 = r0 + rpb; if (r1 > P) r1 = P;
    const float* xb = x + (long long)n * P * ldx + 4 * q;
    float* yb = y + (long long)n * P * ldy + 4 * q;
    for (long long r = r0 + rl; r < r1; r += 4 * RL) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + (long long)u * RL;
            v[u] = rr < r1 ? *reinterpret_cast<const float4*>(xb + rr * ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long rr = r + (long long)u * RL;
            if (rr >= r1) continue;
            float o[4] = {(v[u].x - mu[0]) * rs[0] * ga[0] + be[0], (v[u].y - mu[1]) * rs[1] * ga[1] + be[1],
                          (v[u].z - mu[2]) * rs[2] * ga[2] + be[2], (v[u].w - mu[3]) * rs[3] * ga[3] + be[3]};
            if (relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
            }
            *reinterpret_cast<float4*>(yb + rr * ldy) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// coef[n][g] = (sum_c gamma_c S1, sum_c gamma_c S2) / m ; dgamma_c = sum_n S2 ; dbeta_c = sum_n S1
__global__ void gn_bwd_coef_kernel(const double* __restrict__ chan, const float* __restrict__ gamma, float* __restrict__ coef,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int G, long long P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = C / G;
    if (i < N * G) {
        const int n = i / G, g = i - n * G;
        double a = 0.0, b = 0.0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            a += (double)gamma[c] * chan[2 * (n * C + c)];
            b += (double)gamma[c] * chan[2 * (n * C + c) + 1];
        }
        const double m = (double)P * cg;
        coef[2 * i] = (float)(a / m); coef[2 * i + 1] = (float)(b / m);
    } else if (i < N * G + C) {
        const int c = i - N * G;
        double a = 0.0, b = 0.0;
        for (int n = 0; n < N; ++n) { a += chan[2 * (n * C + c)]; b += chan[2 * (n * C + c) + 1]; }
        dbeta[c] = (float)a; dgamma[c] = (float)b;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ dx, int lddx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ coef,
                                                           long long P, int C, int G, int rpb, int relu,
                                                           const float* __restrict__ add, int ldadd) {      // add (may be NULL): a second gradient of x, summed in here
    const int C4 = C >> 2, RL = 256 / C4, cg = C / G;
    const int tid = threadIdx.x, q = tid % C4, rl = tid / C4, n = blockIdx.y;
    float be[4], rs[4], mu[4], c1[4], c2[4], ga[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * q + k, g = c / cg;
        rs[k] = rstd[n * G + g]; mu[k] = mean[n * G + g]; ga[k] = gamma[c]; be[k] = beta[c];
        c1[k] = coef[2 * (n * G + g)]; c2[k] = coef[2 * (n * G + g) + 1];
    }
    const long long r0 = (long long)blockIdx.x * rpb;
    long long r1 = r0 + rpb; if (r1 > P) r1 = P;
    const float* xb = x + (long long)n * P * ldx + 4 * q;
    const float* dyb = dy + (long long)n * P * lddy + 4 * q;
    float* dxb = dx + (long long)n * P * lddx + 4 * q;
    const float* adb = add ? add + (long long)n * P * ldadd + 4 * q : nullptr;
    for (long long r = r0 + rl; r < r1; r += 2 * RL) {
        float4 v[2], d[2], ad[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long rr = r + (long long)u * RL;
            v[u] = rr < r1 ? *reinterpret_cast<const float4*>(xb + rr * ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
            d[u] = rr < r1 ? *reinterpret_cast<const float4*>(dyb + rr * lddy) : make_float4(0.f, 0.f, 0.f, 0.f);
            ad[u] = (adb && rr < r1) ? *reinterpret_cast<const float4*>(adb + rr * ldadd) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long rr = r + (long long)u * RL;
            if (rr >= r1) continue;
            const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, dv[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xv[k] - mu[k]) * rs[k];
                const float dz = (relu && !(xh * ga[k] + be[k] > 0.f)) ? 0.f : dv[k];
                o[k] = rs[k] * (dz * ga[k] - c1[k] - xh * c2[k]);
            }
            if (adb) {
                // the sum must round exactly as autograd's separate add does: hipcc (-ffp-contract=fast) would fuse the multiply above with this add into an
                // FMA (one rounding instead of two; __fadd_rn is contracted as well) -- the empty asm makes the product opaque first
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(o[k]));
                *reinterpret_cast<float4*>(dxb + rr * lddx) = make_float4(o[0] + ad[u].x, o[1] + ad[u].y, o[2] + ad[u].z, o[3] + ad[u].w);
            } else {
                *reinterpret_cast<float4*>(dxb + rr * lddx) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

static bool gn_ok(int C, int G, int ldx, int ldy) {
    const int C4 = C / 4;
    return C > 0 && G > 0 && C % G == 0 && C % 4 == 0 && C4 <= 256 && 256 % C4 == 0 && ldx % 4 == 0 && ldy % 4 == 0;
}
// rows per block (a multiple of the rows one pass of a block covers) for ~4096 blocks per launch
static int gn_row_blocks(long long P, int C, int N, int unroll, int* rpb) {
    const int pass = (256 / (C / 4)) * unroll;
    long long want = 4096 / (N > 0 ? N : 1); if (want < 1) want = 1;
    long long r = (P + want - 1) / want;
    r = ((r + pass - 1) / pass) * pass;
    if (r > 0x3fffffff) r = 0x3fffffff / pass * pass;
    *rpb = (int)r;
    return (int)((P + r - 1) / r);
}
static int gn_chunks(long long P) { return mrdis_cdiv(P, GN_CHUNK_ROWS); }
static int ew_blocks(long long total) { long long b = (total + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

// workspace: partial (N*nchunk*C*2 floats) | chan (N*C*2 doubles) | coef (N*G*2 floats)
extern "C" size_t mrdis_groupnorm_workspace(int N, long long P, int C, int G) {
    if (N <= 0 || P <= 0 || C <= 0 || G <= 0) return 0;
    size_t part = sizeof(float) * (size_t)N * gn_chunks(P) * C * 2;
    part = (part + 15) & ~(size_t)15;
    return part + sizeof(double) * (size_t)N * C * 2 + sizeof(float) * (size_t)N * G * 2 + 64;
}

extern "C" int mrdis_groupnorm_relu_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                        float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                                        int N, long long P, int C, int G, float eps, int relu, void* stream) {
    if (!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace || N <= 0 || P <= 0) return MRDIS_EINVAL;
    if (!gn_ok(C, G, ldx, ldy) || ldx < C || ldy < C) return MRDIS_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace) & 15) != 0) return MRDIS_EALIGN;
    if (workspace_bytes < mrdis_groupnorm_workspace(N, P, C, G)) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = gn_chunks(P);
    float* partial = reinterpret_cast<float*>(workspace);
    size_t part = sizeof(float) * (size_t)N * nchunk * C * 2; part = (part + 15) & ~(size_t)15;
    double* chan = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + part);
    MRDIS_LAUNCH(gn_partial_kernel<0>, dim3(nchunk, N), dim3(256), 0, s, x, ldx, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                       partial, P, C, G, 0);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(gn_chan_kernel, dim3(N * C), dim3(64), 0, s, partial, chan, N, nchunk, C);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(gn_stat_kernel, dim3(mrdis_cdiv((long long)N * G, 64)), dim3(64), 0, s, chan, save_mean, save_rstd, N, C, G, P, eps);
    MRDIS_CHECK_LAUNCH();
    int rpb; const int nb = gn_row_blocks(P, C, N, 4, &rpb);
    MRDIS_LAUNCH(gn_apply_kernel, dim3(nb, N), dim3(256), 0, s, x, ldx, y, ldy, gamma, beta,
                       save_mean, save_rstd, P, C, G, rpb, relu ? 1 : 0);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_groupnorm_relu_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                                        const float* save_mean, const float* save_rstd, float* dx, int lddx,
                                        float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                        int N, long long P, int C, int G, int relu, void* stream) {
    return mrdis_groupnorm_relu_bwd_add(dy, lddy, x, ldx, gamma, beta, save_mean, save_rstd, dx, lddx, dgamma, dbeta, nullptr, 0, workspace, workspace_bytes,
                                        N, P, C, G, relu, stream);
}

extern "C" int mrdis_groupnorm_relu_bwd_add(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                                            const float* save_mean, const float* save_rstd, float* dx, int lddx,
                                            float* dgamma, float* dbeta, const float* add, int ldadd, void* workspace, size_t workspace_bytes,
                                            int N, long long P, int C, int G, int relu, void* stream) {
    if (!dy || !x || !gamma || !beta || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !workspace || N <= 0 || P <= 0)
        return MRDIS_EINVAL;
    if (!gn_ok(C, G, ldx, lddx) || lddy % 4 != 0 || ldx < C || lddx < C || lddy < C) return MRDIS_EUNSUPPORTED;
    if (add && (ldadd % 4 != 0 || ldadd < C)) return MRDIS_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)workspace | (uintptr_t)add) & 15) != 0) return MRDIS_EALIGN;
    if (workspace_bytes < mrdis_groupnorm_workspace(N, P, C, G)) return MRDIS_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = gn_chunks(P);
    float* partial = reinterpret_cast<float*>(workspace);
    size_t part = sizeof(float) * (size_t)N * nchunk * C * 2; part = (part + 15) & ~(size_t)15;
    double* chan = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + part);
    float* coef = reinterpret_cast<float*>(chan + (size_t)N * C * 2);
    MRDIS_LAUNCH(gn_partial_kernel<1>, dim3(nchunk, N), dim3(256), 0, s, x, ldx, dy, lddy, gamma, beta, save_mean, save_rstd,
                       partial, P, C, G, relu ? 1 : 0);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(gn_chan_kernel, dim3(N * C), dim3(64), 0, s, partial, chan, N, nchunk, C);
    MRDIS_CHECK_LAUNCH();
    MRDIS_LAUNCH(gn_bwd_coef_kernel, dim3(mrdis_cdiv((long long)N * G + C, 64)), dim3(64), 0, s, chan, gamma, coef, dgamma, dbeta, N, C, G, P);
    MRDIS_CHECK_LAUNCH();
    int rpb; const int nb = gn_row_blocks(P, C, N, 2, &rpb);
    MRDIS_LAUNCH(gn_bwd_apply_kernel, dim3(nb, N), dim3(256), 0, s, dy, lddy, x, ldx, dx, lddx,
                       gamma, beta, save_mean, save_rstd, coef, P, C, G, rpb, relu ? 1 : 0, add, ldadd);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

// ---------------------------------------------------------------- nearest x2 upsampling (+ skip), nn.Upsample(scale_factor=2)
// y (N, 2D, 2H, 2W, C) = x (N, D, H, W, C)[d/2, h/2, w/2] + skip ; all contiguous NDHWC, C % 4 == 0
__global__ __launch_bounds__(256) void up2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ skip, float* __restrict__ y,
                                                      int N, int D, int H, int W, int C4) {
    const long long total = (long long)N * 8 * D * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        long long r = i / C4;
        const int w = (int)(r % (2 * W)); r /= 2 * W;
        const int h = (int)(r % (2 * H)); r /= 2 * H;
        const int d = (int)(r % (2 * D));
        const int n = (int)(r / (2 * D));
        const long long src = ((((long long)n * D + (d >> 1)) * H + (h >> 1)) * W + (w >> 1)) * C4 + q;
        float4 v = reinterpret_cast<const float4*>(x)[src];
        if (skip) {
            const float4 k = reinterpret_cast<const float4*>(skip)[i];
            v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
        }
        reinterpret_cast<float4*>(y)[i] = v;
    }
}

__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int D, int H, int W, int C4) {
    const long long total = (long long)N * D * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int q = (int)(i % C4);
        long long r = i / C4;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D);
        const int n = (int)(r / D);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {      // fixed order (dz, dy, dx): matches a sequential sum
            const long long src = ((((long long)n * 2 * D + 2 * d + (k >> 2)) * 2 * H + 2 * h + ((k >> 1) & 1)) * 2 * W + 2 * w + (k & 1)) * C4 + q;
            const float4 v = reinterpret_cast<const float4*>(dy)[src];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(dx)[i] = a;
    }
}

extern "C" int mrdis_upsample2x_add_fwd(const float* x, const float* skip, float* y, int N, int D, int H, int W, int C, void* stream) {
    if (!x || !y || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MRDIS_EINVAL;
    if (C % 4 != 0) return MRDIS_EUNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)skip) & 15) != 0) return MRDIS_EALIGN;
    MRDIS_LAUNCH(up2_fwd_kernel, dim3(ew_blocks((long long)N * 8 * D * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       x, skip, y, N, D, H, W, C / 4);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

extern "C" int mrdis_upsample2x_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, void* stream) {
    if (!dy || !dx || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return MRDIS_EINVAL;
    if (C % 4 != 0) return MRDIS_EUNSUPPORTED;
    if ((((uintptr_t)dy | (uintptr_t)dx) & 15) != 0) return MRDIS_EALIGN;
    MRDIS_LAUNCH(up2_bwd_kernel, dim3(ew_blocks((long long)N * D * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dy, dx, N, D, H, W, C / 4);
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}
