"""bwgrad3_kernel (LDS-DMA, round 5) vs bwgrad2_kernel (register staging; option debug_mode 3010) at the bf16 step's weight-gradient shapes.
   python tools/bwgrad3_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

dev = torch.device('cuda:0')
hip.load()
print('weight gradient (bf16 views)     | bwgrad2 us | bwgrad3 us | TF/s (bwgrad3) | bit-identical')
for (B, ci, co, H, W) in [(32, 128, 256, 64, 64), (32, 64, 128, 128, 128), (32, 32, 64, 256, 256), (32, 128, 64, 64, 64), (32, 64, 32, 128, 128), (32, 128, 128, 32, 32),
                          (32, 256, 128, 32, 32), (32, 512, 128, 32, 32), (128, 128, 256, 32, 32)]:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16)
    dy = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16)
    t, o = {}, {}
    for name, md in (('bwgrad2', 3010), ('bwgrad3', -1), ('bwgrad2 again', 3010), ('bwgrad3 again', -1)):
        hip.set_option('debug_mode', md)
        t[name] = timeit(lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True), iters=20)
        o[name] = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True)
    hip.set_option('debug_mode', -1)
    fl = 2.0 * B * H * W * ci * co * 9
    same = torch.equal(o['bwgrad2'][0], o['bwgrad3'][0]) and torch.equal(o['bwgrad2'][1], o['bwgrad3'][1])
    print(f'{B}x{ci}->{co} {H}x{W}'.ljust(32) + f' | {t["bwgrad2"]:7.1f} / {t["bwgrad2 again"]:7.1f} | {t["bwgrad3"]:7.1f} / {t["bwgrad3 again"]:7.1f} | {fl / t["bwgrad3 again"] / 1e6:8.1f} | {same}', flush=True)
