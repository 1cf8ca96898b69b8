"""Timing-only ablations of wino2_kernel (library built with -DWINO2_ABLATIONS; option debug_mode selects the variant)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
hip.set_option('wino', 2)
for (B, ci, co, H, W) in [(32, 128, 256, 64, 64), (32, 32, 64, 256, 256)]:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(9, ci, co, device=dev) * 0.05
    bias = torch.randn(co, device=dev)
    out = []
    for abl, name in ((-1, 'full'), (3, 'noUV'), (8, 'noFiltLoads'), (32, 'noRawLoads'), (40, 'noGlobal'),
                      (43, 'MFMA+opreads only'), (256, 'no stores'),  (-1, 'full again')):
        hip.set_option('debug_mode', abl)
        out.append(f'{name} {timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1)):.1f}')
    hip.set_option('debug_mode', -1)
    hip.set_option('wino_pipe', 0)
    out.append(f'phase-kernel {timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1)):.1f}')
    hip.set_option('wino_pipe', 1)
    print(f'{B}x{ci}->{co} {H}x{W}: ' + ' | '.join(out), flush=True)
