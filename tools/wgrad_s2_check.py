"""Timing of the stride-2 first-layer weight gradients (mrdis_wgrad_s2.hip) against the generic split-K kernel (option now16 = 1),
at the bench geometry: x = one modality's 7 channels of the (32, 28, 256, 256) batch tensor."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
full = torch.randn(32, 28, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
dense = full[:, 7:14].contiguous(memory_format=torch.channels_last)
for (co, k, x) in [(32, 4, full[:, 7:14]), (16, 3, full[:, 7:14]), (32, 4, dense), (16, 3, dense)]:
    dy = torch.randn(32, co, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
    out = []
    for now16 in (0, 1, 0):
        hip.set_option('debug_now16', now16)
        out.append(f'now16={now16}: {timeit(lambda: hip.conv2d_bwd_weight(x, dy, k, k, 2, 1, need_bias=True)):.1f} us')
    w_tck = torch.randn(k * k, 7, co, device=dev) * 0.1; bias = torch.randn(co, device=dev)
    for now16 in (0, 1):
        hip.set_option('debug_now16', now16)
        out.append(f'fwd now16={now16}: {timeit(lambda: hip.conv2d_fwd(x, w_tck, bias, k, k, 2, 1)):.1f} us')
    w_tkc = torch.randn(k * k, co, 7, device=dev) * 0.1
    for now16 in (0, 1):
        hip.set_option('debug_now16', now16)
        out.append(f'dgrad now16={now16}: {timeit(lambda: hip.conv2d_bwd_data(dy, w_tkc, (256, 256), k, k, 2, 1)):.1f} us')
    hip.set_option('debug_now16', 0)
    mb = (32 * 256 * 256 * 7 + 32 * 128 * 128 * co) * 4 / 1e6
    print(f'7 -> {co} k{k} s2 256x256 B=32 ldx={hip.nhwc(x)[1]} ({mb:.0f} MB algorithmic): ' + ' | '.join(out), flush=True)
