"""Launch ONE entry point of bench.roofline_step (a dominant kernel of the step at the step's shape) a few times:
the target of the PMC (FETCH_SIZE / WRITE_SIZE) passes whose result bench.py reports as `roofline_step.layers[..].<entry>.traffic`.

    python tools/step_kernel.py sp6.gamma+beta fwd_spade [B H W [dtype]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mrdis  # noqa: E402

if __name__ == '__main__':
    layer, entry = sys.argv[1], sys.argv[2]
    B, H, W = (int(v) for v in sys.argv[3:6]) if len(sys.argv) >= 6 else (32, 256, 256)
    dtype = sys.argv[6] if len(sys.argv) > 6 else 'f32'
    r = bench.roofline_step(mrdis, torch.device('cuda:0'), B, H, W, dtype, iters=6, only=(layer, entry))
    print(r['layers'])
