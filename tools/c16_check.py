"""Timing of sp6.out (32 -> 16, 3x3, 256x256, B = 32): mrdis_c16.hip against tapconv16_kernel (option debug_now16 = 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
for (B, ci, H) in [(32, 32, 256), (32, 16, 256), (128, 32, 256)]:
    x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w_tck = torch.randn(9, ci, 16, device=dev) * 0.1; bias = torch.randn(16, device=dev)
    gf = 2 * B * H * H * ci * 16 * 9 / 1e9
    out = []
    for now16 in (0, 1, 0):
        hip.set_option('debug_now16', now16)
        t = timeit(lambda: hip.conv2d_fwd(x, w_tck, bias, 3, 3, 1, 1))
        out.append(f'now16={now16}: {t:.1f} us ({gf / t * 1e3:.0f} TF/s)')
    hip.set_option('debug_now16', 0)
    print(f'B={B} {ci}->16 {H}x{H} ({gf:.1f} GF): ' + ' | '.join(out), flush=True)
