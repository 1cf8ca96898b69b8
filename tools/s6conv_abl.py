"""Tile configurations of the six-product tap kernel (mrdis_s6conv.hip) on a few bench layers, forward and data gradient: microseconds per configuration."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.s6conv_check import cl, tck, tkc, timeit  # noqa: E402,F401

hip = mrdis.hip
dev = torch.device('cuda:0')
B = 32
LAYERS = [('ana.down_2', 32, 64, 4, 2, 1, 128), ('ana.down_3', 64, 128, 4, 2, 1, 64), ('ana.down_4', 128, 256, 4, 2, 1, 32), ('mod.conv3', 32, 64, 3, 2, 1, 64),
          ('sp2.out', 128, 128, 3, 1, 1, 16)]
CFGS = [('auto', -1, -1)] + [(f'{kc}.{wp}x{wc}', 100 * kc + 10 * wp + wc, -1) for kc in (8, 16, 32) for (wp, wc) in ((1, 2), (2, 2), (2, 1), (1, 1))] + [(f'1621g{g}', 1621, g) for g in (32, 64, 128, 256)]
print(f'{"layer":12s} pass  ' + ' '.join(f'{c[0]:>7s}' for c in CFGS) + '   fp32   (us; debug_mode = 100 kc + 10 wp + wc forces the instantiation, 1621gN also the workgroups per cout tile; 0.0 = declined)')
for name, Ci, Co, k, st, pad, H in LAYERS:
    x = torch.randn(B, Ci, H, H); w = torch.randn(Co, Ci, k, k) * 0.05; b = torch.randn(Co)
    Ho = (H + 2 * pad - k) // st + 1
    dy = torch.randn(B, Co, Ho, Ho)
    xd, dyd, wt, wk, bd = cl(x), cl(dy), tck(w), tkc(w), b.to(dev)
    imf, imd = hip.s6_filter_image(wt), hip.s6_filter_image(wk)
    for what, fn in (('fwd', lambda: hip.conv2d_fwd(xd, wt, bd, k, k, st, pad, w_wino=imf)), ('dgrad', lambda: hip.conv2d_bwd_data(dyd, wk, (H, H), k, k, st, pad, w_wino=imd))):
        ts = []
        for _, md, gs in CFGS:
            with hip.option('split6', 10), hip.option('debug_mode', md), hip.option('debug_wgsplit', gs):
                hip.launch_counts(reset=True)
                fn()
                took = hip.launch_counts()['split6_tap'] > 0
                ts.append(timeit(fn) if took else 0.0)
        with hip.option('split6', 0):
            t0 = timeit((lambda: hip.conv2d_fwd(xd, wt, bd, k, k, st, pad)) if what == 'fwd' else (lambda: hip.conv2d_bwd_data(dyd, wk, (H, H), k, k, st, pad)))
        print(f'{name:12s} {what:5s} ' + ' '.join(f'{t:7.1f}' for t in ts) + f'  {t0:6.1f}', flush=True)
