"""Random geometries through the six-product tap-table kernel (mrdis_s6conv.hip; split6 = 10, filter images from mrdis_s6_filter_image) against torch
float64: forward (+ bias, leaky ReLU) and data gradient, filter extents 2..4, strides 1 | 2, paddings 0..2, ragged maps and channel counts, random forced
wave tiles / channel chunks (debug_mode), channel-slice views.  Prints the worst error per pass; exit code 1 on a mismatch (bar: 2e-6 of the maximum, x sqrt(reduction length / 1024) beyond 1,024 terms, or 1.5 x the error of the fp32 MFMA kernels on the same operands).
    python tools/fuzz_s6conv.py [trials]"""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
random.seed(23); torch.manual_seed(23)
TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
BAR = 2e-6


def cl(t):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def rel(a, b):
    return float((a.detach().double().cpu() - b.double().cpu()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


worst = {'fwd': 0.0, 'fwd+lrelu': 0.0, 'dgrad': 0.0}
worst0 = dict(worst)
took = {'fwd': 0, 'dgrad': 0}
ok = True
for trial in range(TRIALS):
    k = random.choice([2, 3, 3, 4, 4]); st = random.choice([1, 2]); pad = random.choice([0, 1, 1, 2 if k > 2 else 1])
    Ci = 8 * random.randint(1, 20); Co = 4 * random.randint(4, 40)
    N = random.choice([1, 2, 3, 5, 8]); H = random.randint(k + 2, 70); W = random.randint(k + 2, 90)
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    if Ho < 1 or Wo < 1:
        continue
    x = torch.randn(N, Ci, H, W); w = torch.randn(Co, Ci, k, k) * (Ci * k * k) ** -0.5; b = torch.randn(Co) * 0.1
    dy = torch.randn(N, Co, Ho, Wo)
    wt = w.permute(2, 3, 1, 0).reshape(-1, Ci, Co).contiguous().to(dev); wk = w.permute(2, 3, 0, 1).reshape(-1, Co, Ci).contiguous().to(dev)
    imf, imd = hip.s6_filter_image(wt), hip.s6_filter_image(wk)
    ref_y = F.conv2d(x.double(), w.double(), b.double(), st, pad)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), w.double(), dy.double(), st, pad)
    xd = cl(x)
    if random.random() < 0.3:                          # a channel slice of a wider buffer
        big = cl(torch.randn(N, Ci + 8, H, W)); big[:, 4:4 + Ci] = x.to(dev); xd = big[:, 4:4 + Ci]
    mode = random.choice([-1] * 12 + [811, 812, 821, 822, 1611, 1612, 1621, 1622, 3211, 3212, 3221, 3222])
    with hip.option('split6', 10), hip.option('debug_mode', mode):
        hip.launch_counts(reset=True)
        y = hip.conv2d_fwd(xd, wt, b.to(dev), k, k, st, pad, w_wino=imf)
        yl = hip.conv2d_fwd(xd, wt, b.to(dev), k, k, st, pad, lrelu=True, w_wino=imf)
        took['fwd'] += hip.launch_counts()['split6_tap'] > 0
        hip.launch_counts(reset=True)
        dx = hip.conv2d_bwd_data(cl(dy), wk, (H, W), k, k, st, pad, w_wino=imd)
        took['dgrad'] += hip.launch_counts()['split6_tap'] > 0
    with hip.option('split6', 0):                      # the fp32 MFMA kernels on the same operands: fp32 accumulation of k k Ci terms has an error of its own
        y0 = hip.conv2d_fwd(xd, wt, b.to(dev), k, k, st, pad)
        dx0 = hip.conv2d_bwd_data(cl(dy), wk, (H, W), k, k, st, pad)
    e0 = {'fwd': rel(y0, ref_y), 'fwd+lrelu': rel(y0, ref_y), 'dgrad': rel(dx0, ref_dx)}
    e = {'fwd': rel(y, ref_y), 'fwd+lrelu': rel(yl, F.leaky_relu(ref_y, 0.2)), 'dgrad': rel(dx, ref_dx)}
    for n_, v in e.items():
        worst[n_] = max(worst[n_], v)
        worst0[n_] = max(worst0[n_], e0[n_])
        if not v <= max(BAR * max(1.0, (k * k * (Ci if n_ != 'dgrad' else Co) / 1024.0) ** 0.5), 1.5 * e0[n_]):      # (rounding of an fp32 accumulation grows like the root of the reduction length)
            ok = False
            print('MISMATCH', n_, v, dict(N=N, Ci=Ci, Co=Co, k=k, stride=st, pad=pad, H=H, W=W, mode=mode), flush=True)
print(f'{TRIALS} trials; the six-product kernel took {took["fwd"]} forward and {took["dgrad"]} data-gradient launches (the rest fell to the fp32 kernels: tile does not fit, reduction not a multiple of 8)')
for n_, v in worst.items():
    print(f'  worst {n_:10s} {v:.2e} of the maximum (bar: {BAR:.0e} x max(1, sqrt(reduction length / 1024)), or 1.5 x the fp32 MFMA kernels\' own error on the same operands; their worst: {worst0[n_]:.2e})')
print('OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
