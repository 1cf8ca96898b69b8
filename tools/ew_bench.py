"""Time the streaming (non-conv) kernels at the bench shapes and report achieved HBM bandwidth
(algorithmic bytes / time): bilinear fwd/bwd, SPADE fwd/bwd, BatchNorm fwd/bwd.  EW_DTYPE=bf16: bf16 activation views (bytes counted at 2 per element)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


T = torch.bfloat16 if os.environ.get('EW_DTYPE', 'f32') == 'bf16' else torch.float32
ES = 2 if T is torch.bfloat16 else 4


def cl(*shape):
    return torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last).to(T)


def main():
    B = 32
    for (C, h) in ((32, 128), (64, 64), (128, 32)):
        x = cl(B, C, h, h); gy = cl(B, C, 2 * h, 2 * h)
        mb_small, mb_big = x.numel() * ES / 1e6, gy.numel() * ES / 1e6
        t = timeit(lambda: hip.bilinear_fwd(x, (2 * h, 2 * h), False))
        print(f'bilinear_fwd  C={C:3d} {h}->{2*h}: {t:7.1f} us  {(mb_small + mb_big) / t:6.2f} TB/s')
        t = timeit(lambda: hip.bilinear_bwd(gy, (h, h), False))
        print(f'bilinear_bwd  C={C:3d} {2*h}->{h}: {t:7.1f} us  {(mb_small + mb_big) / t:6.2f} TB/s')
    for (C, h) in ((32, 256), (64, 128), (128, 64)):
        z = cl(B, C, h, h); gb = cl(B, 2 * C, h, h); go = cl(B, C, h, h)
        mb = z.numel() * ES / 1e6
        out = hip.instnorm_spade_fwd(z, gb[:, :C], gb[:, C:], 1e-5)
        t = timeit(lambda: hip.instnorm_spade_fwd(z, gb[:, :C], gb[:, C:], 1e-5))
        print(f'spade_fwd(+stats) C={C:3d} {h}x{h}: {t:7.1f} us  {5 * mb / t:6.2f} TB/s (5 passes of {mb:.0f} MB)')
        _, mean, rstd = out
        t = timeit(lambda: hip.instnorm_spade_bwd(go, z, gb[:, :C], mean, rstd, fused_gb=True))
        print(f'spade_bwd(+stats) C={C:3d} {h}x{h}: {t:7.1f} us  {9 * mb / t:6.2f} TB/s (9 passes)')
        t2 = timeit(lambda: hip.bilinear_bwd(hip.instnorm_spade_bwd(go, z, gb[:, :C], mean, rstd, fused_gb=True)[0], (h // 2, h // 2), False))
        t3 = timeit(lambda: hip.instnorm_spade_bwd(go, z, gb[:, :C], mean, rstd, fused_gb=True, up2=True))
        xl = cl(B, C, h // 2, h // 2)
        t4 = timeit(lambda: hip.instnorm_spade_bwd(go, None, gb[:, :C], mean, rstd, fused_gb=True, up2=True, xlo=xl))
        print(f'spade_bwd(+stats) + x2 resize adjoint C={C:3d} {h}x{h}: two kernels {t2:7.1f} us, one kernel {t3:7.1f} us, z interpolated from x {t4:7.1f} us')


if __name__ == '__main__':
    main()
