#!/usr/bin/env python3
"""In-process A/B of tapconv variants (debug env switches) on a few layer shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis
hip = mrdis.hip
dev = torch.device('cuda:0')
SHAPES = [('sp6.gamma', 32, 32, 3, 1, 256), ('sp5.gamma', 64, 64, 3, 1, 128), ('sp4.gamma', 128, 128, 3, 1, 64),
          ('up_1.dgrad-like', 32, 128, 3, 1, 128), ('up_2', 256, 64, 3, 1, 64), ('sp6.out', 32, 16, 3, 1, 256)]
MODES = [('nopf16', 1, 0, 16), ('nopf8', 1, 0, 8), ('nopf4', 1, 0, 4), ('pf16', 0, 0, 16), ('pf8', 0, 0, 8)]
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, ci, co, k, s, hw in SHAPES:
    x = torch.randn(32, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(k * k, ci, co, device=dev) * 0.05
    b = torch.zeros(co, device=dev)
    res = []
    for rep in range(1):
        for mname, nopf, leg, kc in MODES:
            os.environ['MRDIS_DEBUG_NOPF'] = str(nopf); os.environ['MRDIS_DEBUG_LEGACY'] = str(leg); os.environ['MRDIS_DEBUG_KC'] = str(kc)
            res.append((mname, t(lambda: hip.conv2d_fwd(x, w, b, k, k, s, 1 if k == 3 else 0))))
    flop = 2.0 * k * k * ci * co * 32 * hw * hw
    print(f'{name:16s}', ' '.join(f'{m}={u:6.1f}({flop / u / 1e6:5.1f})' for m, u in res))
