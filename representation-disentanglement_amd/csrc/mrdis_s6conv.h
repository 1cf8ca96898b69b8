// mrdis_s6conv.hip: the tap-table launch of mrdis_conv.hip on the bf16 matrix pipe with both fp32 operands as three bf16 terms (six products, option split6)
#pragma once
#include "mrdis_tapconv.h"

#define MRDIS_S6_IMAGE_FMT 6          // w_wino_fmt of the convolution entry points: the pointer is a filter image from mrdis_s6_filter_image (include/mrdis.h)

struct S6ConvGeom {
    int tiles;            // position tiles (tilesA * tilesB * tilesN)
    int nchunks;          // Cin / KC
    int tile_stride;      // workgroups walking one cout tile (gridDim.x / coTiles)
    int TinWp;            // padded row pitch (pixels) of the LDS input image
    int groups, taps_img; // 32-cout groups and taps of the filter image
    unsigned long long* stamps; int cap_stamps;      // diagnostic build (-DS6T_STAMPS) only
};
// a planned (not yet launched) launch: the four parity classes of a stride-2 data gradient are planned one by one and launched together
struct S6ConvLaunch { TapConvParams p; S6ConvGeom g; int KC, wp, wc, xr, wr, grid; size_t lds; bool set; };

// image: mrdis_s6_filter_image of the launch's filter ([taps_img][p.Cin][p.Cout] fp32, p.widx[] indexes its taps).  MRDIS_EUNSUPPORTED = not eligible
// (the caller falls back to the fp32 MFMA kernels)
int mrdis_run_s6conv(TapConvParams p, const void* image, int taps_img, int dh_max, int dw_max, hipStream_t s, S6ConvLaunch* defer = nullptr);
int mrdis_launch_s6conv_planned(const S6ConvLaunch (&L)[4], hipStream_t s);
