"""North-star Cin = 4 kernel: persistent-grid size sweep (option c4_grid = workgroups per CU; 0 = from the occupancy query), six-product and fp32 forms,
240x240 (rotating buffers, the roofline shape) and 256x256 (the in-step shape)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mrdis  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
hip.load()
g = torch.Generator().manual_seed(1)
for HW in (240, 256):
    w_cpu = torch.randn(32, 4, 3, 3, generator=g) * 0.1
    w = w_cpu.permute(2, 3, 1, 0).reshape(9, 4, 32).contiguous().to(dev); b = torch.zeros(32, device=dev)
    xs = [torch.randn(32, 4, HW, HW, generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    ys = [hip.empty_nhwc(32, 32, HW, HW, dev) for _ in range(4)]
    for s6 in (1, 0):
        for grid in (0, 2, 3, 4, 5, 6, 8):
            with hip.option('split6', s6), hip.option('c4_grid', grid):
                for _ in range(2):
                    us = bench._time_conv(hip, xs, w, b, ys, 48)
                print(f'{HW}x{HW} split6={s6} c4_grid={grid}: {us:7.2f} us  blocks {hip.get_option("debug_c4_blocks")}', flush=True)
    del xs, ys
