// Shared between mrdis_wino.hip (plan, phase-by-phase kernels) and mrdis_wino2.hip (software-pipelined kernels).
#pragma once
#include "mrdis_common.h"

struct WinoWgradParams {
    const float* x; const float* dy; float* slab; float* bias_slab;
    int N, H, W, Ci, ldx, Co, lddy;
    int nby, nbx, nblocks;        // tile blocks (2 x 4 tiles) per image row / column, total
    int nCiB, nCoB, splits;
    int D, kd;                    // hybrid 3-D form: N counts planes (samples x D); x is read from plane + kd - 1 of the same sample
};

// pipelined 64 x 64 (ci, co) weight-gradient kernel; MRDIS_EUNSUPPORTED -> the caller launches wino_wgrad_kernel<4, 2>
int mrdis_launch_wino_wgrad2(const WinoWgradParams& p, hipStream_t s);

// Winograd F(3x3, 4x4) weight gradient (mrdis_wino4w.hip) on the same plan; sets p.splits to the slabs it wrote; MRDIS_EUNSUPPORTED -> the F(2x2) kernels
int mrdis_launch_wino4_wgrad(WinoWgradParams& p, int max_splits, hipStream_t s);
