"""Timing of sp6.out's weight gradient (32 -> 16, 3x3, 256x256, B = 32): wgrad16_kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
for (B, ci) in [(32, 32), (32, 16)]:
    x = torch.randn(B, ci, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, 16, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    gf = 2 * B * 256 * 256 * ci * 16 * 9 / 1e9
    t = timeit(lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True))
    print(f'B={B} {ci}->16 wgrad ({gf:.1f} GF): {t:.1f} us ({gf / t * 1e3:.0f} TF/s)', flush=True)
