"""bconv4_kernel (LDS-DMA, mrdis_bf16q.hip) vs bconv3_kernel (mrdis_bf16p.hip) at the bf16 step's shapes: forward, data gradient, SPADE-fused form.
   python tools/bconv4_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

dev = torch.device('cuda:0')
B16 = torch.bfloat16
hip.load()
print('layer                          | bconv3 us | bconv4 (policy) | bconv4 (forced) | TF/s (forced) | bit-identical')
for (B, ci, co, H, W, spade) in [(32, 128, 256, 64, 64, 1), (32, 64, 128, 128, 128, 1), (32, 32, 64, 256, 256, 1), (32, 128, 64, 64, 64, 0), (32, 64, 32, 128, 128, 0),
                                 (32, 32, 16, 256, 256, 0), (32, 64, 64, 128, 128, 0), (32, 128, 128, 32, 32, 0), (32, 256, 128, 32, 32, 0), (128, 32, 64, 256, 256, 1)]:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    wt = torch.randn(9, ci, co, device=dev) * 0.05
    wk = wt.permute(0, 2, 1).contiguous()
    wb = hip.cast_bf16(wk)
    bias = torch.randn(co, device=dev)
    t, outs = {}, {}
    for mode in (0, 1, 2):                                # bconv3 | bconv4 where the policy takes it | bconv4 wherever it applies
        hip.set_option('bconv4', mode)
        t[mode] = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb), iters=20)
        outs[mode] = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb)
    fl = 2.0 * B * H * W * ci * co * 9
    hip.set_option('bconv4', 1); hip.set_option('debug_mode', 3003)
    t8 = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb), iters=20)
    hip.set_option('debug_mode', -1)
    t4 = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb), iters=20)
    print(f'      DMA from all 8 waves: {t8:.1f} us | from waves 0-3: {t4:.1f} us')
    d4 = float((outs[2].float() - outs[0].float()).abs().max()) / float(outs[0].float().abs().max())
    print(f'fwd   {B}x{ci}->{co} {H}x{W}'.ljust(30) + f' | {t[0]:9.1f} | {t[1]:9.1f} | {t[2]:9.1f} | {fl / t[2] / 1e6:8.1f} | {torch.equal(outs[0], outs[1])} | forced vs bconv3: max diff {d4:.1e}, '
          f'{float((outs[2] != outs[0]).float().mean()):.1e} of the values differ', flush=True)
    if spade and co % 2 == 0:
        C = co // 2
        z = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
        for mode in (0, 1, 2):
            hip.set_option('bconv4', mode)
            t[mode] = timeit(lambda: hip.gb_spade_fwd(x, wt, bias, z, 1e-5, w_bf16=wb, stats_ready=False), iters=20)
            outs[mode] = hip.gb_spade_fwd(x, wt, bias, z, 1e-5, w_bf16=wb)
        d4 = float((outs[2][0].float() - outs[0][0].float()).abs().max()) / float(outs[0][0].float().abs().max())
        print(f'spade {B}x{ci}->2x{C} {H}x{W} (+stats)'.ljust(30) + f' | {t[0]:9.1f} | {t[1]:9.1f} | {t[2]:9.1f} | {fl / t[2] / 1e6:8.1f} | {all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))} | max diff {d4:.1e}', flush=True)
    del x
hip.set_option('bconv4', 1)
