"""Launch only the north-star 3x3 conv forward (x 32x4x240x240 -> 32 ch, fp32 NHWC) a few times:
the target of the PMC (FETCH_SIZE / WRITE_SIZE) passes whose result bench.py reports as
`roofline.traffic`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mrdis  # noqa: E402

if __name__ == '__main__':
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    r = bench.roofline_conv(mrdis, torch.device('cuda:0'), iters=iters, extras=False)       # the 240x240 shape only (rotating buffers)
    print(r)
