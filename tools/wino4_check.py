"""Winograd F(4x4,3x3) (csrc/mrdis_wino4.hip, option wino4) against the direct kernel (wino = 0) and the pipelined F(2x2,3x3) kernel with its
filter image (wino4 = 0): max difference relative to the maximum of the direct result, and time per call, forward and data gradient.
Also checks the 36-point filter image against G g G^T written out in torch.

    python tools/wino4_check.py [small]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def images(wt, wk, dev):
    """the (forward, data-gradient) filter images in whatever format the library's policy gives the two roles"""
    ci, co = wt.shape[1], wt.shape[2]
    jobs, imgs, blocks = [], [], 0
    for src, R, S, flip in ((wt, ci, co, 0), (wk, co, ci, 1)):
        if S <= 32 and hip.wino_u_format(R, S) != 5:         # (<= 32 couts: only the narrow F(4x4) form reads an image)
            imgs.append(None); continue
        img = torch.zeros(hip.wino_u_image_floats(R, S), device=dev)
        j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC = src.data_ptr(), img.data_ptr(), R, S, flip, 0
        j.block0, j.nblk = blocks, hip.wino_u_job_blocks(R, S); blocks += j.nblk
        jobs.append(j); imgs.append(img)
    if jobs:
        hip.wino_u_jobs(hip.wino_u_table(jobs, dev), len(jobs), blocks)
    return imgs


G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)


def image_reference(w_trs, flip):
    """[cout tile][chunk][18][4][128] image of w [9][R][S] from G g G^T in float64"""
    T, R, S = w_trs.shape
    g = w_trs.double().cpu().reshape(3, 3, R, S)
    if flip:
        g = g.flip(0, 1)
    U = torch.einsum('ai,ijrs,bj->abrs', G4, g, G4).reshape(36, R, S)
    tiles, nch = (S + 63) // 64, (R + 3) // 4
    img = torch.zeros(tiles, nch, 18, 4, 128, dtype=torch.float64)
    for pt in range(36):
        for kq in range(4):
            for m in range(64):
                slot = (2 * m + (pt & 1) + 32 * kq) & 127
                rr = torch.arange(nch) * 4 + kq
                cc = torch.arange(tiles) * 64 + m
                ok_r, ok_c = rr < R, cc < S
                vals = torch.zeros(tiles, nch, dtype=torch.float64)
                vals[ok_c.nonzero()[:, 0][:, None], ok_r.nonzero()[:, 0][None, :]] = U[pt][rr[ok_r]][:, cc[ok_c]].t()
                img[:, :, pt >> 1, kq, slot] = vals
    return img.reshape(-1)


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    shapes = [(2, 64, 32, 40, 37), (32, 64, 32, 256, 256), (32, 64, 32, 128, 128), (32, 128, 32, 128, 128), (2, 32, 64, 20, 37), (3, 40, 72, 50, 70), (1, 64, 128, 64, 64), (32, 32, 64, 256, 256), (32, 64, 128, 128, 128), (32, 128, 256, 64, 64),
              (32, 128, 64, 64, 64), (32, 128, 256, 32, 32), (32, 128, 128, 32, 32), (8, 512, 128, 32, 32), (8, 256, 64, 64, 64), (32, 64, 64, 128, 128)]
    small = len(sys.argv) > 1 and sys.argv[1] == 'small'
    if small:
        shapes = [shapes[0]] + shapes[4:7]
    ok = True
    for (B, ci, co, H, W) in shapes:
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(9, ci, co, device=dev) * 0.05
        wk = wt.permute(0, 2, 1).contiguous()
        bias = torch.randn(co, device=dev)
        dy = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        res = {}
        for name, opts in (('direct', {'wino': 0, 'wino4': 0}), ('f2', {'wino': 2, 'wino4': 0}), ('f4', {'wino': 2, 'wino4': 2})):
            for k, v in opts.items():
                hip.set_option(k, v)
            fmts = (hip.wino_u_format(ci, co), hip.wino_u_format(co, ci))
            im_f, im_b = images(wt, wk, dev) if name != 'direct' else (None, None)
            if name == 'f4' and fmts[0] == 4 and ci * co <= 64 * 128:       # (format 5 is checked through the convolution)
                want = image_reference(wt, 0).float()
                got = im_f.cpu()[:want.numel()]             # (the 16-point fallback image follows)
                e = float((got - want).abs().max() / want.abs().max())
                print(f'   image {ci}->{co}: max |diff| / max = {e:.1e}', flush=True)
                ok = ok and e < 1e-6
            y = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True, w_wino=im_f)
            g = hip.conv2d_bwd_data(dy, wk, (H, W), 3, 3, 1, 1, w_wino=im_b)
            tf = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True, w_wino=im_f)) if not small else 0.0
            td = timeit(lambda: hip.conv2d_bwd_data(dy, wk, (H, W), 3, 3, 1, 1, w_wino=im_b)) if not small else 0.0
            res[name] = (y, g, tf, td, fmts)
        hip.set_option('wino', 1); hip.set_option('wino4', 1)
        yd, gd = res['direct'][0], res['direct'][1]
        line = f'{B}x{ci}->{co} {H}x{W}:'
        for name in ('f2', 'f4'):
            y, g, tf, td, fmts = res[name]
            ey = ((y - yd).abs().max() / yd.abs().max()).item(); eg = ((g - gd).abs().max() / gd.abs().max()).item()
            line += f'  {name}{fmts}: fwd {tf:7.1f} us dgrad {td:7.1f} us (err {ey:.1e} {eg:.1e})'
            ok = ok and ey < 1e-4 and eg < 1e-4
        line += f'  direct: {res["direct"][2]:7.1f} {res["direct"][3]:7.1f}'
        print(line, flush=True)
    print('OK' if ok else 'FAILED', flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
