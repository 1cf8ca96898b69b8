"""A/B of the software-pipelined Winograd kernel (option wino_pipe = 1, mrdis_wino2.hip) against the phase-by-phase one
(wino_pipe = 0) and the direct kernel (wino = 0): max relative difference and time per call, forward and data gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    shapes = [(2, 8, 40, 23, 37), (3, 36, 72, 50, 18), (32, 32, 64, 256, 256), (32, 64, 128, 128, 128), (32, 128, 256, 64, 64),
              (32, 128, 64, 64, 64), (32, 128, 256, 32, 32), (8, 512, 128, 32, 32), (8, 256, 64, 64, 64), (32, 64, 64, 128, 128)]
    if len(sys.argv) > 1 and sys.argv[1] == 'small':
        shapes = shapes[:2]
    for (B, ci, co, H, W) in shapes:
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(9, ci, co, device=dev) * 0.05
        wk = wt.permute(0, 2, 1).contiguous()
        bias = torch.randn(co, device=dev)
        dy = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        res = {}
        for name, opts in (('direct', {'wino': 0}), ('phase', {'wino': 2, 'wino_pipe': 0}), ('pipe', {'wino': 2, 'wino_pipe': 1})):
            for k, v in opts.items():
                hip.set_option(k, v)
            y = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True)
            g = hip.conv2d_bwd_data(dy, wk, (H, W), 3, 3, 1, 1)
            tf = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True))
            td = timeit(lambda: hip.conv2d_bwd_data(dy, wk, (H, W), 3, 3, 1, 1))
            res[name] = (y, g, tf, td)
        hip.set_option('wino', 1); hip.set_option('wino_pipe', 1)
        yd, gd = res['direct'][0], res['direct'][1]
        line = f'{B}x{ci}->{co} {H}x{W}:'
        for name in ('phase', 'pipe'):
            y, g, tf, td = res[name]
            ey = ((y - yd).abs().max() / yd.abs().max()).item(); eg = ((g - gd).abs().max() / gd.abs().max()).item()
            line += f'  {name}: fwd {tf:7.1f} us dgrad {td:7.1f} us (err {ey:.1e} {eg:.1e})'
        line += f'  direct: {res["direct"][2]:7.1f} {res["direct"][3]:7.1f}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
