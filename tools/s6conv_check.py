"""The six-product tap-table kernel (mrdis_s6conv.hip; option split6 = 10: only that kernel) against torch float64 and against the fp32 MFMA kernels
(split6 = 0) on the layers it takes in the bench step -- forward and data gradient, results and time.  Prints per layer:
    error of both forms relative to the result's maximum, microseconds of both, the launch counter that proves which kernel ran.
  python tools/s6conv_check.py [quick]        exit code 1 on a mismatch (bar 2e-6, the bar of the other six-product kernels)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
torch.manual_seed(5)
QUICK = len(sys.argv) > 1 and sys.argv[1] == 'quick'
B = 4 if QUICK else 32


def cl(t):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def tck(w):      # (Co, Ci, kh, kw) -> [taps][Ci][Co]
    return w.permute(2, 3, 1, 0).reshape(-1, w.shape[1], w.shape[0]).contiguous().to(dev)


def tkc(w):      # (Co, Ci, kh, kw) -> [taps][Co][Ci]
    return w.permute(2, 3, 0, 1).reshape(-1, w.shape[0], w.shape[1]).contiguous().to(dev)


def rel(a, b):
    return float((a.detach().double().cpu() - b.double().cpu()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


def timeit(fn, iters=20, batches=3):
    """microseconds per call: the fastest of `batches` batches of `iters` back-to-back calls (a batch now and then catches a multi-millisecond stall of the box)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = float('inf')
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


LAYERS = [      # name, Ci, Co, k, stride, pad, H (square maps)
    ('ana.down_2', 32, 64, 4, 2, 1, 128), ('ana.down_3', 64, 128, 4, 2, 1, 64), ('ana.down_4', 128, 256, 4, 2, 1, 32), ('ana.down_5', 256, 256, 4, 2, 1, 16),
    ('mod.conv2', 16, 32, 3, 2, 1, 128), ('mod.conv3', 32, 64, 3, 2, 1, 64), ('mod.conv4', 64, 128, 3, 2, 1, 32), ('mod.conv5', 128, 128, 3, 2, 1, 16),
    ('sp2.gamma+beta', 128, 256, 3, 1, 1, 16), ('sp1.gamma+beta', 128, 256, 3, 1, 1, 8), ('sp2.out', 128, 128, 3, 1, 1, 16), ('sp1.out', 128, 128, 3, 1, 1, 8),
    ('ana.up_4', 256, 256, 3, 1, 1, 16), ('sp3.out', 128, 128, 3, 1, 1, 32), ('sp3.gamma+beta', 128, 256, 3, 1, 1, 32), ('odd', 24, 20, 3, 1, 1, 19),
    ('odd s2', 40, 36, 4, 2, 1, 22),
]


def main():
    global LAYERS
    if QUICK:
        LAYERS = [l for l in LAYERS if l[6] <= 64]
    ok = True
    print(f'{"layer":16s} {"Ci":>4s} {"Co":>4s} k s {"HxW":>8s} | fwd err s6 / fp32    us s6 / fp32 | dgrad err s6 / fp32    us s6 / fp32 | kernels (fwd, dgrad)')
    for name, Ci, Co, k, st, pad, H in LAYERS:
        x = torch.randn(B, Ci, H, H); w = torch.randn(Co, Ci, k, k) * (1.0 / (Ci * k * k) ** 0.5); b = torch.randn(Co) * 0.1
        Ho = (H + 2 * pad - k) // st + 1
        dy = torch.randn(B, Co, Ho, Ho)
        xd, dyd, wt, wk, bd = cl(x), cl(dy), tck(w), tkc(w), b.to(dev)
        imf, imd = hip.s6_filter_image(wt), hip.s6_filter_image(wk)
        ref_y = F.conv2d(x.double().to(dev), w.double().to(dev), b.double().to(dev), st, pad)
        ref_dx = torch.nn.grad.conv2d_input((B, Ci, H, H), w.double().to(dev), dy.double().to(dev), st, pad)
        res = {}
        for mode in (10, 0):
            with hip.option('split6', mode):
                imf_, imd_ = (imf, imd) if mode == 10 else (None, None)
                hip.launch_counts(reset=True)
                y = hip.conv2d_fwd(xd, wt, bd, k, k, st, pad, w_wino=imf_)
                cf = hip.launch_counts()['split6_tap']
                hip.launch_counts(reset=True)
                dx = hip.conv2d_bwd_data(dyd, wk, (H, H), k, k, st, pad, w_wino=imd_)
                cd = hip.launch_counts()['split6_tap']
                tf = timeit(lambda: hip.conv2d_fwd(xd, wt, bd, k, k, st, pad, w_wino=imf_))
                td = timeit(lambda: hip.conv2d_bwd_data(dyd, wk, (H, H), k, k, st, pad, w_wino=imd_))
            res[mode] = (rel(y, ref_y), rel(dx, ref_dx), tf, td, cf, cd)
        a, f = res[10], res[0]
        took = a[4] > 0 or a[5] > 0
        bad = took and (not a[0] <= max(2e-6, 2.0 * f[0]) or not a[1] <= max(2e-6, 2.0 * f[1]))      # (K = 16 taps x 256 channels: fp32 accumulation itself is at 1.6e-6)
        ok = ok and not bad
        print(f'{name:16s} {Ci:4d} {Co:4d} {k} {st} {H:4d}x{H:<4d}| {a[0]:.1e} / {f[0]:.1e}  {a[2]:7.1f} / {f[2]:7.1f} | {a[1]:.1e} / {f[1]:.1e}  {a[3]:7.1f} / {f[3]:7.1f} | '
              f's6 launches {a[4]}, {a[5]}' + ('   MISMATCH' if bad else ''), flush=True)
    print('OK' if ok else 'FAILED')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
