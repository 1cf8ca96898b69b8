"""Timing-only ablations of spade_bwd_up2_kernel<T, ONEPASS, 512> (ab/libmrdis_abl_elem.so, built -DSPADE_UP2_ABL; results wrong): what bounds the kernel.
   bits: 1 no streaming loads | 2 no z interpolation (xt reads) | 4 no dgamma / dbeta stores | 8 no phase 2 (the adjoint from LDS)
   python tools/spade_up2_abl.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402

dev = torch.device('cuda:0')
hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl_elem.so'))


def timed(fn, reps=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


for dt in (torch.float32, torch.bfloat16):
    for (N, C, Hi) in [(128, 32, 128), (128, 64, 64)]:
        H = W = 2 * Hi
        x = cl(torch.randn(N, C, Hi, Hi, device=dev)).to(dt)
        gb = cl(torch.randn(N, 2 * C, H, W, device=dev)).to(dt)
        gamma = gb[:, :C]
        dout = cl(torch.randn(N, C, H, W, device=dev)).to(dt)
        mean = torch.zeros(N * C, device=dev); rstd = torch.ones(N * C, device=dev)
        row = []
        for abl in (0, 1, 2, 4, 8, 6, 14, 15, 0):
            hip.set_option('debug_mode', 2100 + abl if abl else -1)
            row.append((abl, timed(lambda: hip.instnorm_spade_bwd(dout, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x))))
        hip.set_option('debug_mode', -1)
        print(f'{str(dt)[6:]:9s} N={N} C={C} {H}x{W} (three launches: this kernel + stat_final + the low-resolution finish): ' + ' | '.join(f'abl {a}: {t:.0f} us' for a, t in row), flush=True)
