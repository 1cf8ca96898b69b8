"""A/B of the fused gamma | beta + SPADE-modulation convolution (mrdis_conv2d_fwd_spade): F(2x2,3x3) kernel (wino4 = 0) vs F(4x4,3x3) kernel (wino4 = 2)
at the SPADE blocks of the benchmarked step.   python tools/spade4_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

if len(sys.argv) > 1:
    mrdis.hip.load(sys.argv[1])                       # another build of the library (A/B)
from tools.wino4_check import timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
for (N, Ci, C, H) in [(32, 32, 32, 256), (32, 64, 64, 128), (32, 128, 128, 64), (128, 128, 128, 32), (128, 128, 128, 16)]:
    x = torch.randn(N, Ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    z = torch.randn(N, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(9, Ci, 2 * C, device=dev) * 0.05
    b = torch.randn(2 * C, device=dev)
    out = []
    ref = None
    for name, w4 in (('F2', 0), ('F4', 2)):
        hip.set_option('wino', 2); hip.set_option('wino4', w4)
        img = torch.zeros(hip.wino_u_image_floats(Ci, 2 * C, C), device=dev)
        j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC, j.block0, j.nblk = wt.data_ptr(), img.data_ptr(), Ci, 2 * C, 0, C, 0, hip.wino_u_job_blocks(Ci, 2 * C, C)
        hip.wino_u_jobs(hip.wino_u_table([j], dev), 1, j.nblk)
        res = hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img)
        if res is None:
            out.append(f'{name} declined'); continue
        bufs = res
        t = timeit(lambda: hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img, out=bufs, stats_ready=True))
        if w4:
            hip.set_option('debug_mode', 2005)          # F(4x4): without the z prefetch
            t_np = timeit(lambda: hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img, out=bufs, stats_ready=True))
            r_np = hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img)
            hip.set_option('debug_mode', -1)
            t2 = timeit(lambda: hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img, out=bufs, stats_ready=True))
            assert torch.equal(r_np[0], res[0]) and torch.equal(r_np[1], res[1])
            ts = []
            for md in (2006, 2008, 2007, -1):          # stores: plain | gamma non-temporal | both non-temporal | default (gamma nt, mix nt when the tensor is >= nt_mb)
                hip.set_option('debug_mode', md)
                ts.append(timeit(lambda: hip.gb_spade_fwd(x, wt, b, z, 1e-5, w_wino=img, out=bufs, stats_ready=True)))
            hip.set_option('debug_mode', -1)
            name = f'{name} [no z prefetch {t_np:.1f} us, again with {t2:.1f} us; stores plain / gamma nt / both nt / default: ' + ' / '.join(f'{v:.1f}' for v in ts) + ']'
        if ref is None:
            ref = res[0].clone()
            out.append(f'{name} fmt {hip.wino_u_format(Ci, 2 * C, C)}: {t:.1f} us')
        else:
            out.append(f'{name} fmt {hip.wino_u_format(Ci, 2 * C, C)}: {t:.1f} us (max diff vs F2 {float((res[0] - ref).abs().max() / ref.abs().max()):.1e})')
    hip.set_option('wino', 1); hip.set_option('wino4', 1)
    print(f'{N}x{Ci}->2x{C} {H}x{H}: ' + ' | '.join(out), flush=True)
