"""Timing-only ablations of bconv4_kernel (ab/libmrdis_abl_bf16.so: mrdis_bf16q.hip built -DBCONV4_ABLATIONS; option debug_mode selects; results of the
ablated variants are wrong by construction).   bash tools/build_abl.sh && python tools/bconv4_abl.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

mrdis.hip.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ab', 'libmrdis_abl_bf16.so'))
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
B16 = torch.bfloat16
for (B, ci, co, H, W) in [(32, 128, 256, 64, 64), (32, 64, 128, 128, 128), (32, 32, 64, 256, 256)]:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    wt = torch.randn(9, ci, co, device=dev) * 0.05
    wb = hip.cast_bf16(wt.permute(0, 2, 1).contiguous())
    bias = torch.randn(co, device=dev)
    out = []
    for abl, name in ((-1, 'full'), (1, 'noMFMA'), (2, 'noDMA'), (4, 'noOutStores'), (8, 'noDMAwait'), (6, 'noDMA+noStores (MFMA + LDS reads + barriers)'), (10, 'noDMA+noWait'),
                      (14, 'noDMA noStores noWait'), (7, 'skeleton: LDS reads + barriers + epilogue math'), (-1, 'full again')):
        hip.set_option('debug_mode', abl)
        out.append(f'{name} {timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb), iters=20):.1f}')
    hip.set_option('debug_mode', -1)
    print(f'{B}x{ci}->{co} {H}x{W}: ' + ' | '.join(out), flush=True)
