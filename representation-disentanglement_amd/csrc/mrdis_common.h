// Shared helpers for libmrdis_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mrdis.h"

#define MRDIS_CHECK_LAUNCH()                                     \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) return MRDIS_ELAUNCH;              \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Process-wide switches.  Each is read from its environment variable ONCE (first use) and afterwards only changes through
// mrdis_set_option(): no getenv() on the launch path.  Boolean debug switches: variable present = 1.  Value switches: -1 = unset.
enum {
    MRDIS_OPT_WINO, MRDIS_OPT_NT_MB, MRDIS_OPT_WINO_PIPE, MRDIS_OPT_WINO_U, MRDIS_OPT_WINO4, MRDIS_OPT_WINO4R, MRDIS_OPT_BCONV4, MRDIS_OPT_SPLIT6,
    MRDIS_OPT_NO16, MRDIS_OPT_NOTHIN, MRDIS_OPT_NOC4, MRDIS_OPT_NODMA, MRDIS_OPT_NO16_3D, MRDIS_OPT_BILGEN, MRDIS_OPT_NOW16, MRDIS_OPT_NOPACK,
    MRDIS_OPT_MODE, MRDIS_OPT_BN, MRDIS_OPT_KC, MRDIS_OPT_BM, MRDIS_OPT_C4_TW, MRDIS_OPT_WGSPLIT, MRDIS_OPT_BN3, MRDIS_OPT_KC3,
    MRDIS_OPT_C4_GRID, MRDIS_OPT_C4_BLOCKS,
    MRDIS_OPT_COUNT
};
long long mrdis_opt(int id);      // mrdis_elem.hip
void mrdis_opt_note(int id, long long value);      // diagnostics a launcher leaves behind (read with mrdis_get_option)

// Launch counters of the Winograd / bf16 / six-product (split6) kernel families (host side, one increment per launch): what a test asks to know which form actually ran
// (mrdis_launch_count("wino4") ...; mrdis_elem.hip).
enum { MRDIS_CNT_WINO, MRDIS_CNT_WINO_SPADE, MRDIS_CNT_WINO2, MRDIS_CNT_WINO2_SPADE, MRDIS_CNT_WINO4, MRDIS_CNT_WINO4_SPADE, MRDIS_CNT_WINO4N, MRDIS_CNT_WINO4R,
       MRDIS_CNT_WINO_WGRAD, MRDIS_CNT_WINO_WGRAD2, MRDIS_CNT_WINO4_WGRAD, MRDIS_CNT_BCONV3, MRDIS_CNT_BCONV3_SPADE, MRDIS_CNT_BCONV4, MRDIS_CNT_BCONV4_SPADE,
       MRDIS_CNT_SPLIT6_C4, MRDIS_CNT_SPLIT6_C16, MRDIS_CNT_SPLIT6_WGRAD16, MRDIS_CNT_SPLIT6_CO4, MRDIS_CNT_SPLIT6_C3D, MRDIS_CNT_SPLIT6_W3D, MRDIS_CNT_SPLIT6_TAP,
       MRDIS_CNT_ALL /* every launch of the library */, MRDIS_CNT_COUNT };
void mrdis_count(int id);

// Every kernel launch of the library goes through MRDIS_LAUNCH: it records, per kernel expression, the largest DYNAMIC LDS size it was launched with
// (rocprofv3's kernel trace reports only the static group segment: 0 for the `extern __shared__` kernels, e.g. the 155 KB of wino4_kernel).
// mrdis_dynamic_lds_table (mrdis_elem.hip) hands the table out; bench.py puts it into its JSON line, tools/prof_summary.py into the per-kernel tables.
void mrdis_note_lds(const char* kernel_expr, size_t bytes);
#define MRDIS_LAUNCH(kernel, grid, block, lds, s, ...) do { mrdis_count(MRDIS_CNT_ALL); if ((size_t)(lds) != 0) mrdis_note_lds(#kernel, (size_t)(lds)); hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__); } while (0)

static inline int mrdis_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Bijective XCD-aware remap (cdna_hip_programming.md T1): workgroups b and b+8 share an XCD, so
// give each XCD a contiguous run of logical tile ids -> neighbouring tiles hit the same L2.
__device__ __forceinline__ int mrdis_xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ float mrdis_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double mrdis_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// mrdis_wgrad_s2.hip: dw[i] = sum_k slab[k * total + i], dbias[co] (+)= sum_k bslab[k * Co + co], fixed order
int mrdis_launch_slab_reduce(const float* slab, float* dw, int total, int Co, int nslab, const float* bslab, float* dbias, int accumulate_bias,
                             hipStream_t s);
