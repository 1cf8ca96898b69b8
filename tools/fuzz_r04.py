"""Random small shapes through the round-4 kernels against their references: mrdis_instnorm_spade_bwd_up2 (one-pass and two-pass forms) vs the kernels it replaces,
wino4r_kernel (both forms) vs torch.  Prints the worst error per kernel; exit code 1 on a mismatch."""
import os, sys, random
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis
from tools.wino4_check import images
hip = mrdis.hip; dev = torch.device('cuda:0')
random.seed(7); torch.manual_seed(7)


def cl(t):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


ok = True
worst = {'up2 dx': 0.0, 'up2 dgb': 0.0, 'wino4r': 0.0}
for trial in range(60):
    N = random.choice([1, 2, 3, 5]); C = random.choice([4, 8, 16, 32, 40, 48, 64, 96, 128]); h = random.randint(1, 21); w = random.randint(1, 21)
    if h * w < 4:                 # (a constant up-sampled map: rstd ~ 300 and d x = 0 up to rounding in either form)
        h, w = 2, 2
    x = cl(torch.randn(N, C, h, w)); z = hip.bilinear_fwd(x, (2 * h, 2 * w), False)
    gamma = cl(torch.randn(N, C, 2 * h, 2 * w) * 0.3); dout = cl(torch.randn(N, C, 2 * h, 2 * w))
    mean = z.mean(dim=(2, 3)).reshape(-1).contiguous(); rstd = (1.0 / (z.var(dim=(2, 3), unbiased=False) + 1e-5).sqrt()).reshape(-1).contiguous()
    dz, dgb_ref = hip.instnorm_spade_bwd(dout, z, gamma, mean, rstd, fused_gb=True)
    dx_ref = hip.bilinear_bwd(dz, (h, w), False)
    for mode in (-1, 2001):
        hip.set_option('debug_mode', mode)
        res = hip.instnorm_spade_bwd(dout, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x)
        hip.set_option('debug_mode', -1)
        if res is None:
            continue
        dx, dgb = res
        scale = max(float(dx_ref.abs().max()), 1e-3 * float(rstd.max()) * float(dout.abs().max()))      # (a one-pixel map: d x is exactly 0 up to rounding of rstd ~ 300 terms)
        ex = float((dx - dx_ref).abs().max()) / scale; eg = float((dgb - dgb_ref).abs().max() / (dgb_ref.abs().max() + 1e-20))
        worst['up2 dx'] = max(worst['up2 dx'], ex); worst['up2 dgb'] = max(worst['up2 dgb'], eg)
        if not (ex < 5e-5 and eg < 5e-6):
            print('MISMATCH up2', (N, C, h, w), mode, ex, eg); ok = False
hip.set_option('wino', 2); hip.set_option('wino4', 2)
for trial in range(40):
    B = random.choice([1, 2, 3]); ci = random.choice([16, 24, 32, 40, 64, 72]); co = random.choice([4, 8, 12, 16, 20, 28, 32])
    H = random.randint(16, 70); W = random.randint(32, 100)
    if hip.wino_u_format(ci, co) != 5:
        continue
    x = torch.randn(B, ci, H, W); wgt = torch.randn(co, ci, 3, 3) * 0.05; b = torch.randn(co) * 0.1
    wt = wgt.permute(2, 3, 1, 0).reshape(9, ci, co).contiguous().to(dev)
    im_f, _ = images(wt, wt.permute(0, 2, 1).contiguous(), dev)
    want = F.leaky_relu(F.conv2d(x, wgt, b, 1, 1), 0.2)
    for mode in (2, 3):
        hip.set_option('wino4r', mode)
        y = hip.conv2d_fwd(cl(x), wt, b.to(dev), 3, 3, 1, 1, lrelu=True, w_wino=im_f)
        e = float((y.cpu() - want).abs().max() / want.abs().max())
        worst['wino4r'] = max(worst['wino4r'], e)
        if not e < 5e-5:
            print('MISMATCH wino4r', (B, ci, co, H, W), mode, e); ok = False
hip.set_option('wino4r', 1); hip.set_option('wino', 1); hip.set_option('wino4', 1)
torch.cuda.synchronize()
print('worst errors:', worst, 'OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
