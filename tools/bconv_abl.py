"""Timing-only ablations of the bf16 convolution kernels (library built with -DBCONV_ABLATIONS -DBCONV3_ABLATIONS; option debug_mode selects)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

mrdis.hip.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ab', 'libmrdis_abl_bf16.so'))   # tools/build_abl.sh
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
B16 = torch.bfloat16
for (B, ci, co, H, W) in [(32, 128, 256, 64, 64), (32, 64, 128, 128, 128), (32, 32, 64, 256, 256)]:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    wt = torch.randn(9, ci, co, device=dev) * 0.05
    wk = wt.permute(0, 2, 1).contiguous()
    wb = hip.cast_bf16(wk)
    bias = torch.randn(co, device=dev)
    out = []
    for abl, name in ((-1, 'full'), (1, 'noMFMA'), (2, 'noGlobalLoads'), (4, 'noLdsStores'), (8, 'opreads same addr'), (16, 'noOutStores'), (6, 'noLoads+noLdsStores'),
                      (14, 'MFMA+epilogue only'), (15, 'epilogue only'), (30, 'loop skeleton'), (-1, 'full again')):
        hip.set_option('debug_mode', abl)
        out.append(f'{name} {timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb)):.1f}')
    hip.set_option('debug_mode', -1)
    print(f'{B}x{ci}->{co} {H}x{W}: ' + ' | '.join(out), flush=True)
