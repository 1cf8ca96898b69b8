// mrdis_conv3d.h -- the tap-table description of one 3-D convolution launch, shared by mrdis_conv3d.hip (generic + 16-cout fp32 MFMA kernels)
// and mrdis_conv3d_s6.hip (the six-product 16 -> 16 kernel).  See the formulation at the top of mrdis_conv3d.hip.
#pragma once
#include "mrdis_common.h"

#define T3_TAPS 27
#define T3_TAB_INTS 320   // tab_in[128] tab_out[128] tap_xoff[32] tap_widx[32]

struct Conv3dParams {
    const float* in; const float* w; const float* bias; const float* res; float* out;
    int N, Din, Hin, Win, Cin, ldin;
    int Dout, Hout, Wout, Cout, ldout, ldres;
    int Z, A, B, os, od0, oh0, ow0, is;
    int ntaps;
    int dd[T3_TAPS], dh[T3_TAPS], dw[T3_TAPS], widx[T3_TAPS];
    int dd_min, dh_min, dw_min;
    int TD, TH, TW, TinD, TinH, TinW;
    int tilesZ, tilesA, tilesB, coTiles;
    int vec_in, vec_w;
};

// mrdis_conv3d_s6.hip: 3x3x3 / stride 1, 16 -> 16 channels as six bf16 products per fp32 product (option split6); MRDIS_EUNSUPPORTED outside that
int mrdis_run_conv3d16_s6(const Conv3dParams& p, long long ptiles_hint, hipStream_t s);
// the same layers' weight (+ bias) gradient as slabs for wgrad3d16_reduce_kernel (CW = 16, nCi = nCo = 1)
int mrdis_run_wgrad3d16_s6(const float* x, int ldx, const float* dy, int lddy, float* slab, size_t slab_bytes, int want_bias,
                           int N, int D, int H, int W, int Ci, int Co, int* splits_out, float** bias_slab_out, hipStream_t s);
