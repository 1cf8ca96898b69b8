"""Timing of the 1x1 decoder head (16 -> 7 at 256x256, B = 32): streaming kernels (mrdis_pointwise.hip) against the generic tile kernels
(option debug_now16 = 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
for B in (32, 128):
    x = torch.randn(B, 16, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, 7, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    w_tck = torch.randn(1, 16, 7, device=dev); w_tkc = w_tck.permute(0, 2, 1).contiguous(); bias = torch.randn(7, device=dev)
    mb = B * 256 * 256 * 23 * 4 / 1e6
    for now16 in (0, 1):
        hip.set_option('debug_now16', now16)
        f = timeit(lambda: hip.conv2d_fwd(x, w_tck, bias, 1, 1, 1, 0))
        d = timeit(lambda: hip.conv2d_bwd_data(dy, w_tkc, (256, 256), 1, 1, 1, 0))
        g = timeit(lambda: hip.conv2d_bwd_weight(x, dy, 1, 1, 1, 0, need_bias=True))
        print(f'B={B} now16={now16} ({mb:.0f} MB per pass): fwd {f:.1f} us ({mb / f:.2f} TB/s) | dgrad {d:.1f} us | wgrad {g:.1f} us', flush=True)
    hip.set_option('debug_now16', 0)
