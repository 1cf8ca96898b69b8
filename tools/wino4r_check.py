"""The <= 32-cout Winograd F(4x4,3x3) forms against the direct kernel: the shared-transform form of mrdis_wino4.hip (option wino4r = 0) and the
register-fed form of mrdis_wino4r.hip with 64-tile workgroups (wino4r = 2) / channel-split wave pairs (wino4r = 3); error relative to the
maximum of the direct result, time per call.

    python tools/wino4r_check.py [small]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino4_check import images, timeit  # noqa: E402

hip = mrdis.hip


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    shapes = [(2, 64, 32, 40, 37), (3, 24, 20, 50, 70), (1, 16, 8, 16, 32), (2, 72, 32, 33, 95), (32, 64, 32, 256, 256), (32, 64, 32, 128, 128), (32, 128, 32, 128, 128),
              (32, 32, 16, 256, 256), (32, 16, 32, 256, 256)]
    small = len(sys.argv) > 1 and sys.argv[1] == 'small'
    if small:
        shapes = shapes[:4]
    ok = True
    for (B, ci, co, H, W) in shapes:
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(9, ci, co, device=dev) * 0.05
        wk = wt.permute(0, 2, 1).contiguous()
        bias = torch.randn(co, device=dev)
        hip.set_option('wino', 0); hip.set_option('wino4', 0)
        yd = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True)
        td = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True)) if not small else 0.0
        hip.set_option('wino', 2); hip.set_option('wino4', 2)
        assert hip.wino_u_format(ci, co) == 5, (ci, co)
        im_f, _ = images(wt, wk, dev)
        line = f'{B}x{ci}->{co} {H}x{W}: direct/F2 {td:7.1f} us'
        for mode in (0, 2, 3):
            hip.set_option('wino4r', mode)
            y = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True, w_wino=im_f)
            t = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True, w_wino=im_f)) if not small else 0.0
            e = ((y - yd).abs().max() / yd.abs().max()).item()
            line += f' | wino4r={mode}: {t:7.1f} us err {e:.1e}'
            ok = ok and e < 1e-4
        print(line, flush=True)
        hip.set_option('wino', 1); hip.set_option('wino4', 1); hip.set_option('wino4r', 1)
    print('OK' if ok else 'FAILED', flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
