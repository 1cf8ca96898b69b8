"""Per-layer timing of the conv entry points at the bench configuration (B=32, M=4, 256x256):
every distinct (Ci, Co, k, stride, H, W) of the hot path, forward / data-gradient /
weight-gradient, HIP events on the launch stream.  Prints a table sorted by the layer's
share of one training step (calls/step from SURVEY.md Appendix A)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip


def layers(B, H, W):
    """(name, Ci, Co, k, stride, pad, Hin, Win, calls_per_step, needs_dgrad_calls)"""
    L = []
    # anatomy encoder (8 calls/step; first-layer dgrad only in pass 2 -> 4 calls)
    c = 32
    L.append(('ana.down_1', 7, c, 4, 2, 1, H, W, 8, 4))
    chans = [(c, 2 * c), (2 * c, 4 * c), (4 * c, 8 * c), (8 * c, 8 * c)]
    h, w = H // 2, W // 2
    for i, (a, b) in enumerate(chans):
        L.append((f'ana.down_{i + 2}', a, b, 4, 2, 1, h, w, 8, 8)); h //= 2; w //= 2
    # anatomy decoder: conv on the x2-upsampled map
    ups = [('up_4', 8 * c, 8 * c, H // 16), ('up_3', 16 * c, 4 * c, H // 8), ('up_2', 8 * c, 2 * c, H // 4),
           ('up_1', 4 * c, c, H // 2), ('output', 2 * c, 4, H)]
    for n, a, b, hh in ups:
        L.append((f'ana.{n}', a, b, 3, 1, 1, hh, hh * W // H, 8, 8))
    # modality encoder
    me = [(7, 16), (16, 32), (32, 64), (64, 128), (128, 128)]
    h, w = H, W
    for i, (a, b) in enumerate(me):
        L.append((f'mod.conv{i + 1}', a, b, 3, 2, 1, h, w, 8, 4 if i == 0 else 8)); h //= 2; w //= 2
    # SPADE blocks (16 decoder calls/step)
    sp = [(128, 128, 32), (128, 128, 16), (128, 128, 8), (128, 64, 4), (64, 32, 2), (32, 16, 1)]
    for i, (ci, co, d) in enumerate(sp):
        hh, ww = H // d, W // d
        L.append((f'sp{i + 1}.si', 4, ci, 3, 1, 1, hh, ww, 16, 16))
        L.append((f'sp{i + 1}.gamma+beta', ci, 2 * ci, 3, 1, 1, hh, ww, 16, 16))      # as the step runs them: ONE fused conv
        L.append((f'sp{i + 1}.out', ci, co, 3, 1, 1, hh, ww, 16, 16))
    L.append(('dec.out1x1', 16, 7, 1, 1, 0, H, W, 16, 16))
    return L


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--hw', type=int, nargs=2, default=[256, 256])
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--no-wino-images', action='store_true', help='fp32: the pipelined Winograd kernels transform the filter taps themselves (as before round 3)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'bf16m'],
                    help='bf16: bf16 activations + bf16 MFMA operands (fp32 kernels between view casts where the bf16 kernels do not apply: '
                         'timed with the casts); bf16m: bf16 MFMA operands on fp32 activations')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    B, (H, W) = a.batch, a.hw
    rows = []
    for name, ci, co, k, s, p, hi, wi, calls, dcalls in layers(B, H, W):
        if a.only and not any(k_ in name for k_ in a.only.split(',')):
            continue
        x = torch.randn(B, ci, hi, wi, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(k * k, ci, co, device=dev) * 0.05
        wk = wt.permute(0, 2, 1).contiguous()
        bias = torch.zeros(co, device=dev)
        ho, wo = hip.conv_out_hw(hi, wi, k, k, s, p)
        dy = torch.randn(B, co, ho, wo, device=dev).contiguous(memory_format=torch.channels_last)
        flop = 2.0 * k * k * ci * co * B * ho * wo
        byts = 4.0 * (B * ci * hi * wi + B * co * ho * wo)
        bf = a.dtype != 'f32'
        wb_f = hip.cast_bf16(wk) if bf else None
        wb_b = hip.cast_bf16(wt) if bf else None
        dt = hip.DT_F32_BF16M if bf else hip.DT_F32
        if a.dtype == 'bf16':
            # storage mode as the step runs it (the launches of ops.conv2d + its autograd adjoint, called directly so that the host does
            # not bound the timing): bf16 MFMA kernels on bf16 views, a 4- / 7-channel side of a stride-1 layer zero-padded to 16 (the
            # padded filter is what the mixing launch delivers; the view casts that pad / slice the activations are timed), or the fp32
            # kernels between view casts on the stride-2 first layers
            import torch.nn.functional as F
            B16 = torch.bfloat16
            xs = x.to(B16) if ci >= 16 else x
            dys = dy.to(B16) if co >= 16 else dy
            byts = xs.element_size() * xs.numel() + dys.element_size() * dys.numel()
            elig = ci % 16 == 0 and co % 4 == 0 and co >= 16
            if k == 1 and ci == 16 and co <= 8:        # the 1x1 head: mixed-storage streaming kernels (bf16 in, fp32 out)
                dxo = hip.empty_nhwc(B, ci, hi, wi, dev, B16)
                f_fwd = lambda: hip.conv2d_fwd(xs, wt, bias, k, k, s, p, out_dtype=torch.float32)
                f_dgrad = lambda: hip.conv2d_bwd_data(dys, wk, (hi, wi), k, k, s, p, out=dxo)
                f_wgrad = lambda: hip.conv2d_bwd_weight(xs, dys, k, k, s, p)
            elif elig or s == 1:
                cip, cop = max(ci, 16), max(co, 16)
                wtp = F.pad(wt, (0, cop - co, 0, cip - ci)); wkp = F.pad(wk, (0, cip - ci, 0, cop - co)); bp = F.pad(bias, (0, cop - co))
                wbf, wbb = hip.cast_bf16(wkp), hip.cast_bf16(wtp)
                xin = hip.cast_view(xs, B16, cip); dyp = hip.cast_view(dys, B16, cop)

                def f_fwd():
                    y = hip.conv2d_fwd(hip.cast_view(xs, B16, cip), wtp, bp, k, k, s, p, w_bf16=wbf)
                    return y if cop == co else hip.cast_view(y, torch.float32, co)
                if ci == 4 and k == 3 and s == 1:        # the Cin = 4 kernels read the fp32 map: bf16 out (forward), bf16 dy (weight gradient); maps
                    c4w = hip.conv2d_bwd_weight(xs, dys, k, k, s, p, may_decline=True) is not None      # narrower than 64 make the zero-padded bf16 copy
                    def f_fwd():                         # noqa: F811
                        if not c4w:
                            hip.cast_view(xs, B16, cip)
                        return hip.conv2d_fwd(xs, wtp, bp, k, k, s, p, out_dtype=B16)

                def f_dgrad():
                    g = hip.conv2d_bwd_data(hip.cast_view(dys, B16, cop), wkp, (hi, wi), k, k, s, p, w_bf16=wbb)
                    return g if cip == ci else hip.cast_view(g, torch.float32, ci)
                if ci == 4 and k == 3 and s == 1:       # si_layers' data gradient: the 4-cout kernel reads the bf16 gradient, fp32 out (maps 64 .. 256 wide)
                    dx4 = hip.empty_nhwc(B, 4, hi, wi, dev, torch.float32)
                    if hip.conv2d_bwd_data(dys, wkp, (hi, wi), k, k, s, p, out=dx4, may_decline=True) is not None:
                        f_dgrad = lambda: hip.conv2d_bwd_data(dys, wkp, (hi, wi), k, k, s, p, out=dx4, may_decline=True)      # noqa: E731
                if co == 4 and k == 3 and s == 1 and ci % 16 == 0:      # C -> 4 (ana_dec.output): forward on the 4-cout kernel (bf16 in, fp32 out)
                    y4 = hip.empty_nhwc(B, 4, ho, wo, dev, torch.float32)
                    if hip.conv2d_fwd(xs, wtp, bp, k, k, s, p, out=y4, may_decline=True) is not None:
                        f_fwd = lambda: hip.conv2d_fwd(xs, wtp, bp, k, k, s, p, out=y4, may_decline=True)      # noqa: E731
                if co == 4 and k == 3 and s == 1 and ci % 16 == 0:      # ... and its data gradient's Cin = 4 kernel reads the fp32 dy, bf16 out
                    dxo4 = hip.empty_nhwc(B, ci, hi, wi, dev, B16)
                    if hip.conv2d_bwd_data(dys, wkp, (hi, wi), k, k, s, p, out=dxo4, may_decline=True) is not None:
                        f_dgrad = lambda: hip.conv2d_bwd_data(dys, wkp, (hi, wi), k, k, s, p, out=dxo4, may_decline=True)      # noqa: E731

                co4w = co == 4 and k == 3 and s == 1 and ci % 16 == 0 and hip.conv2d_bwd_weight(xs, dys, k, k, s, p, may_decline=True) is not None

                def f_wgrad():
                    if co4w:                              # C -> 4: the bf16 trunk x the fp32 gradient (wgrad_c4b_kernel<.., SWAP>), no padded bf16 copy of dy
                        return hip.conv2d_bwd_weight(xs, dys, k, k, s, p, may_decline=True)
                    if ci == 4 and k == 3 and s == 1:
                        return hip.conv2d_bwd_weight(xs, dys, k, k, s, p, may_decline=True) if c4w else hip.conv2d_bwd_weight(xs, hip.cast_view(dys, torch.float32), k, k, s, p)
                    return hip.conv2d_bwd_weight(xin, dyp if cop == co else hip.cast_view(dys, B16, cop), k, k, s, p, dtype=hip.DT_F32_BF16M)
            else:
                x32 = hip.cast_view(xs, torch.float32)

                def f_fwd():
                    y = hip.conv2d_fwd(hip.cast_view(xs, torch.float32), wt, bias, k, k, s, p)
                    return hip.cast_view(y, B16) if co >= 16 else y

                def f_dgrad():
                    g = hip.conv2d_bwd_data(hip.cast_view(dys, torch.float32), wk, (hi, wi), k, k, s, p)
                    return hip.cast_view(g, B16) if ci >= 16 else g

                def f_wgrad():
                    return hip.conv2d_bwd_weight(x32, hip.cast_view(dys, torch.float32), k, k, s, p)
            tf, td, tw = timeit(f_fwd, a.iters), timeit(f_dgrad, a.iters), timeit(f_wgrad, a.iters)
        else:
            if a.dtype == 'f32' and k == 3 and s == 1 and not a.no_wino_images and ci % 4 == 0 and co % 4 == 0:
                # as in the step: the filter's Winograd-domain images, built once behind the mixing launch (mrdis_wino_u_jobs)
                jobs, imgs, blocks = [], [], 0
                for src, R, S, flip in ((wt, ci, co, 0), (wk, co, ci, 1)):
                    if S <= 32:
                        imgs.append(None); continue
                    img = torch.zeros(hip.wino_u_image_floats(R, S), device=dev)
                    j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC = src.data_ptr(), img.data_ptr(), R, S, flip, 0
                    j.block0, j.nblk = blocks, hip.wino_u_job_blocks(R, S); blocks += j.nblk
                    jobs.append(j); imgs.append(img)
                if jobs:
                    hip.wino_u_jobs(hip.wino_u_table(jobs, dev), len(jobs), blocks)
                wb_f, wb_b = imgs
            tf = timeit(lambda: hip.conv2d_fwd(x, wt, bias, k, k, s, p, w_bf16=wb_f), a.iters)
            td = timeit(lambda: hip.conv2d_bwd_data(dy, wk, (hi, wi), k, k, s, p, w_bf16=wb_b), a.iters)
            tw = timeit(lambda: hip.conv2d_bwd_weight(x, dy, k, k, s, p, dtype=dt), a.iters)
        rows.append((name, ci, co, k, s, hi, wi, calls, flop, byts, tf, td, tw, dcalls))
        del x, dy
    tot = sum(r[10] * r[7] + r[11] * r[13] + r[12] * r[7] for r in rows)
    print(f'{"layer":18s} {"Ci":>4s} {"Co":>4s} k s {"HxW":>9s} calls {"GF":>7s} | {"fwd us":>8s} {"TF/s":>6s} {"GB/s":>6s} | {"dgrad us":>8s} {"TF/s":>6s} | '
          f'{"wgrad us":>8s} {"TF/s":>6s} | step ms  share')
    for r in sorted(rows, key=lambda r: -(r[10] * r[7] + r[11] * r[13] + r[12] * r[7])):
        name, ci, co, k, s, hi, wi, calls, flop, byts, tf, td, tw, dcalls = r
        ms = (tf * calls + td * dcalls + tw * calls) / 1e3
        print(f'{name:18s} {ci:4d} {co:4d} {k} {s} {hi:4d}x{wi:<4d} {calls:5d} {flop / 1e9:7.2f} | {tf:8.1f} {flop / tf / 1e6:6.1f} {byts / tf / 1e3:6.0f} | '
              f'{td:8.1f} {flop / td / 1e6:6.1f} | {tw:8.1f} {flop / tw / 1e6:6.1f} | {ms:7.2f} {100 * ms * 1e3 / tot:5.1f}%')
    print(f'conv total per step: {tot / 1e3:.1f} ms')


if __name__ == '__main__':
    main()
