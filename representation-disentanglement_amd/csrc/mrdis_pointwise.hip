// mrdis_pointwise.hip -- the 1x1 decoder head (SPADENewNotShared.out, model.py:2605: 16 -> 7 channels at full resolution, 16 calls per
// step) as three streaming kernels: forward, data gradient, weight (+ bias) gradient.  112 FMAs per pixel against 92 bytes: HBM-bound
// (193 MB per call at B = 32); the generic MFMA tile kernels ran it at 2.4 TB/s (7-float output rows, LDS staging, 16-wide tiles that are
// 56 % padding).  Here a pixel is a QUAD of lanes, lane q owns channels 4q .. 4q+3 of the 16-channel side:
//   forward   every lane loads one float4 (a wave reads 1 KB contiguous), forms 7 partial sums, the quad adds them with two DPP steps,
//             lane q stores outputs q and q + 4: a wave writes 448 contiguous bytes in two instructions
//   dgrad     the quad's lanes read the pixel's 7 dy values (same addresses: one request), each produces its float4 of dx
//   wgrad     28 + 7 running sums per lane over a grid-stride pixel set, lanes of equal q are added at the end (wave shuffles, LDS across
//             waves), one slab row per workgroup, fixed-order slab reduction (mrdis_launch_slab_reduce): deterministic
// The 16-channel side is templated on its storage type T16 (float | __bf16): under `compute_dtype: bf16` the decoder trunk is stored in
// bf16 while the 7-channel reconstruction stays fp32 (MRDIS_DT_XBF16_YF32) -- the head then reads / writes bf16 directly (8 bytes per lane)
// instead of running as a 16 -> 16 bf16 MFMA layer between a zero-padding cast and a slicing cast (70 / 88 / 114 us vs 35 / 33 / 37 us at B = 32).
#include "mrdis_common.h"

namespace {
constexpr int PW_CI = 16, PW_MAXCO = 8;

typedef __bf16 pw_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pw_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 pw_ld4(const __bf16* p) {
    const pw_bf16x4 t = *reinterpret_cast<const pw_bf16x4*>(p);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}
__device__ __forceinline__ void pw_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void pw_st4(__bf16* p, float4 v) {
    pw_bf16x4 t; t[0] = (__bf16)v.x; t[1] = (__bf16)v.y; t[2] = (__bf16)v.z; t[3] = (__bf16)v.w;      // round to nearest even
    *reinterpret_cast<pw_bf16x4*>(p) = t;
}

__device__ __forceinline__ float pw_quad_sum(float v) {
    // v + quad_perm(1,0,3,2)(v), then + quad_perm(2,3,0,1): every lane of the quad ends with the sum of the four
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);
    v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true);
    return v + __int_as_float(t);
}
}  // namespace

// y[pix][co] = bias[co] + sum_ci x[pix][ci] w[ci][co]   (w_tck = [1][16][Co])
template <int CO, typename T16>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const T16* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ y, int ldy, long long npix, int lrelu) {
    const int q = threadIdx.x & 3;
    float wq[4][CO], b0 = 0.f, b1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CO; ++c) wq[j][c] = w[(4 * q + j) * CO + c];
    if (bias != nullptr) { b0 = bias[q]; if (q + 4 < CO) b1 = bias[q + 4]; }
    const long long nitems = npix * 4, gsz = (long long)gridDim.x * blockDim.x;
    for (long long it0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; it0 < nitems; it0 += 4 * gsz) {
        float4 xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long it = it0 + u * gsz;
            xv[u] = it < nitems ? pw_ld4(x + (it >> 2) * ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long it = it0 + u * gsz;
            float s[CO];
#pragma unroll
            for (int c = 0; c < CO; ++c) s[c] = pw_quad_sum(xv[u].x * wq[0][c] + xv[u].y * wq[1][c] + xv[u].z * wq[2][c] + xv[u].w * wq[3][c]);
            float o0 = q == 0 ? s[0] : q == 1 ? s[1 % CO] : q == 2 ? s[2 % CO] : s[3 % CO];
            float o1 = q == 0 ? s[4 % CO] : q == 1 ? s[5 % CO] : q == 2 ? s[6 % CO] : s[7 % CO];
            o0 += b0; o1 += b1;
            if (lrelu) { o0 = o0 > 0.f ? o0 : 0.2f * o0; o1 = o1 > 0.f ? o1 : 0.2f * o1; }
            if (it < nitems) {
                float* d = y + (it >> 2) * ldy;
                if (q < CO) d[q] = o0;
                if (q + 4 < CO) d[q + 4] = o1;
            }
        }
    }
}

// dx[pix][ci] = sum_co dy[pix][co] w[co][ci]   (w_tkc = [1][Co][16])
template <int CO, typename T16>
__global__ __launch_bounds__(256) void pw_dgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                       T16* __restrict__ dx, int lddx, long long npix) {
    const int q = threadIdx.x & 3;
    float wq[CO][4];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[c][j] = w[c * PW_CI + 4 * q + j];
    const long long nitems = npix * 4, gsz = (long long)gridDim.x * blockDim.x;
    for (long long it0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; it0 < nitems; it0 += 2 * gsz) {
        float g[2][CO];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long it = it0 + u * gsz;
            const float* s = dy + (it >> 2) * lddy;
#pragma unroll
            for (int c = 0; c < CO; ++c) g[u][c] = it < nitems ? s[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long it = it0 + u * gsz;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < CO; ++c) { o.x += g[u][c] * wq[c][0]; o.y += g[u][c] * wq[c][1]; o.z += g[u][c] * wq[c][2]; o.w += g[u][c] * wq[c][3]; }
            if (it < nitems) pw_st4(dx + (it >> 2) * lddx + 4 * q, o);
        }
    }
}

// slab[block][ci][co] = sum over the block's pixels of x[pix][ci] dy[pix][co];  bias_slab[block][co] = sum dy[pix][co]
template <int CO, typename T16>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const T16* __restrict__ x, int ldx, const float* __restrict__ dy, int lddy,
                                                       float* __restrict__ slab, float* __restrict__ bias_slab, long long npix) {
    __shared__ float red[4][4][4 * CO + CO];          // [wave][q][4 channels x CO | CO column sums]
    const int tid = threadIdx.x, q = tid & 3, lane = tid & 63, wave = tid >> 6;
    float acc[4][CO], bs[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) { bs[c] = 0.f; acc[0][c] = acc[1][c] = acc[2][c] = acc[3][c] = 0.f; }
    const long long nitems = npix * 4, gsz = (long long)gridDim.x * blockDim.x;
    for (long long it0 = (long long)blockIdx.x * blockDim.x + tid; it0 < nitems; it0 += 2 * gsz) {
        float4 xv[2]; float g[2][CO];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long it = it0 + u * gsz;
            const bool ok = it < nitems;
            xv[u] = ok ? pw_ld4(x + (it >> 2) * ldx + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float* s = dy + (it >> 2) * lddy;
#pragma unroll
            for (int c = 0; c < CO; ++c) g[u][c] = ok ? s[c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                acc[0][c] += xv[u].x * g[u][c]; acc[1][c] += xv[u].y * g[u][c]; acc[2][c] += xv[u].z * g[u][c]; acc[3][c] += xv[u].w * g[u][c];
                bs[c] += g[u][c];
            }
    }
    // lanes of equal q (lane bits 2..5), fixed order
#pragma unroll
    for (int m = 4; m < 64; m <<= 1) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][c] += __shfl_xor(acc[j][c], m, 64);
            bs[c] += __shfl_xor(bs[c], m, 64);
        }
    }
    if (lane < 4) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave][q][j * CO + c] = acc[j][c];
            red[wave][q][4 * CO + c] = bs[c];
        }
    }
    __syncthreads();
    constexpr int TOTAL = PW_CI * CO;
    if (tid < TOTAL) {
        const int ci = tid / CO, c = tid - ci * CO, qq = ci >> 2, j = ci & 3;
        slab[(long long)blockIdx.x * TOTAL + tid] = (red[0][qq][j * CO + c] + red[1][qq][j * CO + c]) + (red[2][qq][j * CO + c] + red[3][qq][j * CO + c]);
    } else if (tid < TOTAL + CO && bias_slab != nullptr) {
        const int c = tid - TOTAL;                    // the column sums of the q = 0 lanes: every pixel exactly once
        bias_slab[(long long)blockIdx.x * CO + c] = (red[0][0][4 * CO + c] + red[1][0][4 * CO + c]) + (red[2][0][4 * CO + c] + red[3][0][4 * CO + c]);
    }
}

// x16_bf16: the 16-channel side is a bf16 view (ld in bf16 elements).  For fp32 views the kernels are a measured policy (large maps only, option
// debug_now16 = 1 turns them off); for the mixed-storage head they are the only kernels, so every size qualifies.
static bool pw_ok(int Ci, int Co, int ld16, const void* p16, long long npix, bool x16_bf16) {
    if (!(Ci == PW_CI && Co >= 1 && Co <= PW_MAXCO && ld16 % 4 == 0 && (((uintptr_t)p16) & (x16_bf16 ? 7 : 15)) == 0)) return false;
    return x16_bf16 ? npix >= 1 : (npix >= 4096 && !mrdis_opt(MRDIS_OPT_NOW16));
}
static int pw_grid(long long nitems, int per_thread) {
    long long g = (nitems + 256LL * per_thread - 1) / (256LL * per_thread);
    if (g > 4096) g = 4096;
    return (int)(g < 1 ? 1 : g);
}
#define PW_BY_CO(K, T, ...) mrdis_count(MRDIS_CNT_ALL); switch (Co) { case 1: K<1, T> __VA_ARGS__; break; case 2: K<2, T> __VA_ARGS__; break; case 3: K<3, T> __VA_ARGS__; break; \
    case 4: K<4, T> __VA_ARGS__; break; case 5: K<5, T> __VA_ARGS__; break; case 6: K<6, T> __VA_ARGS__; break; case 7: K<7, T> __VA_ARGS__; break; default: K<8, T> __VA_ARGS__; break; }

// each returns MRDIS_EUNSUPPORTED outside what it covers (16 channels on the wide side -- fp32 or bf16 views --, <= 8 fp32 channels on the narrow one)
int mrdis_run_pw_fwd(const void* x, int ldx, const float* w_tck, const float* bias, float* y, int ldy, long long npix, int Ci, int Co,
                     int lrelu, int x16_bf16, hipStream_t s) {
    if (!pw_ok(Ci, Co, ldx, x, npix, x16_bf16 != 0)) return MRDIS_EUNSUPPORTED;
    const int grid = pw_grid(npix * 4, 4);
    if (x16_bf16) { PW_BY_CO(pw_fwd_kernel, __bf16, <<<dim3(grid), dim3(256), 0, s>>>((const __bf16*)x, ldx, w_tck, bias, y, ldy, npix, lrelu)) }
    else { PW_BY_CO(pw_fwd_kernel, float, <<<dim3(grid), dim3(256), 0, s>>>((const float*)x, ldx, w_tck, bias, y, ldy, npix, lrelu)) }
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

int mrdis_run_pw_dgrad(const float* dy, int lddy, const float* w_tkc, void* dx, int lddx, long long npix, int Ci, int Co, int x16_bf16, hipStream_t s) {
    if (!pw_ok(Ci, Co, lddx, dx, npix, x16_bf16 != 0)) return MRDIS_EUNSUPPORTED;
    const int grid = pw_grid(npix * 4, 2);
    if (x16_bf16) { PW_BY_CO(pw_dgrad_kernel, __bf16, <<<dim3(grid), dim3(256), 0, s>>>(dy, lddy, w_tkc, (__bf16*)dx, lddx, npix)) }
    else { PW_BY_CO(pw_dgrad_kernel, float, <<<dim3(grid), dim3(256), 0, s>>>(dy, lddy, w_tkc, (float*)dx, lddx, npix)) }
    MRDIS_CHECK_LAUNCH();
    return MRDIS_OK;
}

size_t mrdis_pw_wgrad_workspace(long long npix, int Ci, int Co, int x16_bf16) {
    if (Ci != PW_CI || Co < 1 || Co > PW_MAXCO || (!x16_bf16 && npix < 4096) || npix < 1) return 0;
    return sizeof(float) * (size_t)512 * (PW_CI * Co + Co) + 256;
}

int mrdis_run_pw_wgrad(const void* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                       long long npix, int Ci, int Co, int accumulate_bias, int x16_bf16, hipStream_t s) {
    if (!pw_ok(Ci, Co, ldx, x, npix, x16_bf16 != 0)) return MRDIS_EUNSUPPORTED;
    if (workspace_bytes + 256 < mrdis_pw_wgrad_workspace(npix, Ci, Co, x16_bf16)) return MRDIS_EUNSUPPORTED;
    int grid = pw_grid(npix * 4, 16);
    if (grid > 512) grid = 512;
    float* slab = reinterpret_cast<float*>(workspace);
    float* bslab = dbias ? slab + (size_t)grid * PW_CI * Co : nullptr;
    if (x16_bf16) { PW_BY_CO(pw_wgrad_kernel, __bf16, <<<dim3(grid), dim3(256), 0, s>>>((const __bf16*)x, ldx, dy, lddy, slab, bslab, npix)) }
    else { PW_BY_CO(pw_wgrad_kernel, float, <<<dim3(grid), dim3(256), 0, s>>>((const float*)x, ldx, dy, lddy, slab, bslab, npix)) }
    MRDIS_CHECK_LAUNCH();
    return mrdis_launch_slab_reduce(slab, dw_tck, PW_CI * Co, Co, grid, bslab, dbias, accumulate_bias, s);
}
