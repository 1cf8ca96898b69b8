"""Timing of the 4-cout 3x3 layers (ana_dec.output forward 64 -> 4 at 256x256; si_layers data gradients 32 -> 4 at 256x256, 64 -> 4 at
128x128), B = 32: mrdis_co4.hip against tapconv16_kernel<., THIN4> (option debug_now16 = 1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
for (B, ci, H) in [(32, 64, 256), (32, 32, 256), (32, 64, 128)]:
    x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    w_tck = torch.randn(9, ci, 4, device=dev) * 0.1; bias = torch.randn(4, device=dev)
    w_tkc = torch.randn(9, ci, 4, device=dev) * 0.1
    mb = B * H * H * (ci + 4) * 4 / 1e6
    out = []
    for now16 in (0, 1, 0):
        hip.set_option('debug_now16', now16)
        f = timeit(lambda: hip.conv2d_fwd(x, w_tck, bias, 3, 3, 1, 1))
        d = timeit(lambda: hip.conv2d_bwd_data(x, w_tkc, (H, H), 3, 3, 1, 1))
        out.append(f'now16={now16}: fwd {f:.1f} us ({mb / f:.2f} TB/s) dgrad-of-4->{ci} {d:.1f} us')
    hip.set_option('debug_now16', 0)
    print(f'B={B} {ci}->4 {H}x{H} ({mb:.0f} MB): ' + ' | '.join(out), flush=True)
