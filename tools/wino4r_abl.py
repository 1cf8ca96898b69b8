"""Timing-only variants of wino4r_kernel (ab/libmrdis_abl.so, option debug_mode; results wrong): which resource a stage waits for.
    python tools/wino4r_abl.py [N Ci Co H W]
bits: 1 no patch reads (d = 1), 2 no filter reads, 4 no MFMAs, 40 = 8 | 32 no copies"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

lib = mrdis.hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl.so'))
from tools.wino4_check import images, timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
a = [int(v) for v in sys.argv[1:]]
N, ci, co, H, W = a[:5] if len(a) >= 5 else (32, 64, 32, 128, 128)
hip.set_option('wino', 2); hip.set_option('wino4', 2)
x = torch.randn(N, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(9, ci, co, device=dev) * 0.05
bias = torch.randn(co, device=dev)
im_f, _ = images(wt, wt.permute(0, 2, 1).contiguous(), dev)
names = {0: 'full', 1: 'no patch reads', 2: 'no filter reads', 3: 'no LDS reads', 4: 'no MFMAs', 40: 'no copies', 43: 'MFMAs + transform arithmetic only', 47: 'skeleton (barriers, epilogue)'}
for mode in (2, 3):
    hip.set_option('wino4r', mode)
    line = f'{N}x{ci}->{co} {H}x{W} wino4r={mode}:'
    for abl in (0, 1, 2, 3, 4, 40, 43, 47):
        hip.set_option('debug_mode', abl if abl else -1)
        t = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_wino=im_f), iters=20)
        line += f'  {names[abl]} {t:.1f}'
    hip.set_option('debug_mode', -1)
    print(line, flush=True)
