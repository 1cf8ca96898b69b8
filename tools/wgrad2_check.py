"""A/B of the pipelined Winograd weight-gradient kernel (wino_pipe = 1) against the phase-by-phase one and the direct kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    shapes = [(2, 64, 64, 23, 37), (3, 128, 64, 50, 18), (32, 64, 128, 128, 128), (32, 128, 256, 64, 64), (32, 128, 64, 64, 64),
              (32, 128, 256, 32, 32), (8, 512, 128, 32, 32), (8, 256, 64, 64, 64), (32, 64, 64, 128, 128), (32, 128, 128, 64, 64)]
    if len(sys.argv) > 1 and sys.argv[1] == 'small':
        shapes = shapes[:2]
    for (B, ci, co, H, W) in shapes:
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        res = {}
        for name, opts in (('direct', {'wino': 0}), ('phase', {'wino': 2, 'wino_pipe': 0}), ('pipe', {'wino': 2, 'wino_pipe': 1})):
            for k, v in opts.items():
                hip.set_option(k, v)
            dw, db = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True)
            t = timeit(lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True))
            res[name] = (dw, db, t)
        hip.set_option('wino', 1); hip.set_option('wino_pipe', 1)
        dwd, dbd = res['direct'][0], res['direct'][1]
        line = f'{B}x{ci}->{co} {H}x{W}:'
        for name in ('phase', 'pipe'):
            dw, db, t = res[name]
            e1 = ((dw - dwd).abs().max() / dwd.abs().max()).item(); e2 = ((db - dbd).abs().max() / dbd.abs().max()).item()
            line += f'  {name}: {t:7.1f} us (err dw {e1:.1e} db {e2:.1e})'
        line += f'  direct: {res["direct"][2]:7.1f}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
