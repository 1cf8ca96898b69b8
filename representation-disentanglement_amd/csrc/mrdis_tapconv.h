// Tap-table convolution parameters shared by the fp32 kernels (mrdis_conv.hip) and the bf16-MFMA kernels (mrdis_bf16.hip).
#pragma once
#include "mrdis_common.h"

#define TC_BM 128
struct TapConvParams {
    const float* in; const float* w; const float* bias; float* out;
    int N, Hin, Win, Cin, ldin;
    int Hout, Wout, Cout, ldout;
    int A, B, os, oh0, ow0, is;
    int ntaps;
    int dh[MRDIS_MAX_TAPS], dw[MRDIS_MAX_TAPS], widx[MRDIS_MAX_TAPS];
    int dh_min, dw_min;
    int NB, TH, TW, TinH, TinW;
    int tilesA, tilesB, tilesN, coTiles;
    int epilogue;
    int vec_in, vec_w;
    int prefetch;                 // staging mode of tapconv_kernel: 0 generic | 1 hoisted descriptors + register prefetch
    const void* w_bf16;           // bf16 filter with the REDUCTION axis contiguous, [tap][Cout][Cin] of this launch (or nullptr)
    int dtype;                    // MRDIS_DT_*
    const void* s6_img; int s6_taps;      // six-product filter image of this launch's filter (mrdis_s6_filter_image) and its tap count, or nullptr (mrdis_s6conv.hip)
};


// bf16 kernels (mrdis_bf16.hip): geometry of a launch, and a planned-but-not-launched launch (the four parity classes of a stride-2
// data gradient are planned one by one and launched together)
struct BConvGeom {
    int tiles;            // position tiles (tilesA * tilesB * tilesN)
    int nchunks;          // Cin / KC
    int tile_stride;      // workgroups walking one cout tile (gridDim.x / coTiles)
};
struct BConvLaunch { TapConvParams p; BConvGeom g; int KC, waves_c, wp, wc, grid; size_t lds; bool set; };

struct TileChoice { int NB, TH, TW; };

static inline TileChoice choose_tile(int N, int A, int B, int BMv = TC_BM) {
    TileChoice best{1, 1, 1};
    double best_u = -1.0;
    for (int tw = 1; tw <= 32 && tw <= B; ++tw) {
        int th = BMv / tw; if (th > A) th = A;
        int nb = BMv / (tw * th); if (nb > N) nb = N; if (nb < 1) nb = 1;
        const double u = ((double)B / ((double)mrdis_cdiv(B, tw) * tw)) * ((double)A / ((double)mrdis_cdiv(A, th) * th)) *
                         ((double)N / ((double)mrdis_cdiv(N, nb) * nb)) * ((double)(tw * th * nb) / BMv);
        // prefer wide rows (coalesced staging, conflict-free LDS reads) on ties
        if (u > best_u + 1e-9 || (u > best_u - 1e-9 && tw > best.TW)) { best_u = u; best = {nb, th, tw}; }
    }
    return best;
}


// mrdis_bf16.hip: the same tap-table launch on v_mfma_f32_32x32x16_bf16; MRDIS_EUNSUPPORTED = not eligible (caller falls back)
int mrdis_run_bconv(TapConvParams p, int dh_max, int dw_max, hipStream_t s, BConvLaunch* defer = nullptr);
int mrdis_launch_bconv_planned(const BConvLaunch (&L)[4], hipStream_t s);
