"""Winograd F(3x3,4x4) weight gradient (csrc/mrdis_wino4w.hip, option wino4) against the direct kernels (wino = 0) and the F(2x2) weight-gradient kernels
(wino4 = 0): difference relative to the direct result's norm / maximum, time per call.    python tools/wgrad4_check.py [small]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino4_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
torch.manual_seed(0)
shapes = [(2, 32, 64, 16, 24), (3, 64, 64, 40, 32), (32, 128, 256, 64, 64), (32, 64, 128, 128, 128), (32, 32, 64, 256, 256), (32, 128, 64, 64, 64), (128, 128, 256, 32, 32),
          (32, 256, 64, 64, 64), (32, 512, 128, 32, 32), (128, 128, 128, 32, 32)]
small = len(sys.argv) > 1 and sys.argv[1] == 'small'
if small:
    shapes = shapes[:2]
ok = True
for (B, ci, co, H, W) in shapes:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dy = (torch.randn(B, co, H, W, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    res = {}
    for name, opts in (('direct', {'wino': 0, 'wino4': 0}), ('f2', {'wino': 2, 'wino4': 0}), ('f4', {'wino': 2, 'wino4': 2})):
        for k, v in opts.items():
            hip.set_option(k, v)
        dw, db = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1)
        t = 0.0 if small else timeit(lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1))
        res[name] = (dw.clone(), db.clone(), t)
    hip.set_option('wino', 1); hip.set_option('wino4', 1)
    dwd, dbd = res['direct'][0], res['direct'][1]
    line = f'{B}x{ci}->{co} {H}x{W}:'
    for name in ('f2', 'f4'):
        dw, db, t = res[name]
        en = float((dw - dwd).norm() / dwd.norm()); em = float((dw - dwd).abs().max() / dwd.abs().max()); eb = float((db - dbd).abs().max() / dbd.abs().max())
        line += f'  {name}: {t:7.1f} us (norm err {en:.1e}, max err {em:.1e}, bias {eb:.1e})'
        ok = ok and en < 1e-4 and em < 5e-4 and eb < 1e-4
    line += f'  direct: {res["direct"][2]:7.1f} us;  f4 == f2: {bool(torch.equal(res["f4"][0], res["f2"][0]))}'
    print(line, flush=True)
print('OK' if ok else 'FAILED', flush=True)
sys.exit(0 if ok else 1)
