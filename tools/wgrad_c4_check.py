"""Timing of the si_layers weight gradients (4 -> C, 3x3 s1) at B = 32: wgrad_c4_kernel against wgrad_thin_dma_kernel (option debug_now16 = 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
for (co, H) in [(32, 256), (64, 128), (128, 64)]:
    x = torch.randn(32, 4, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(32, co, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    mb = 32 * H * H * (4 + co) * 4 / 1e6
    out = []
    for now16 in (0, 1, 0):
        hip.set_option('debug_now16', now16)
        t = timeit(lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True))
        out.append(f'now16={now16}: {t:.1f} us ({mb / t:.2f} TB/s)')
    hip.set_option('debug_now16', 0)
    print(f'4->{co} {H}x{H} B=32 ({mb:.0f} MB): ' + ' | '.join(out), flush=True)
