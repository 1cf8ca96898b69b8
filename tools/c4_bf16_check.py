"""c4conv_kernel with bf16 output (MRDIS_DT_XF32_YBF16) against its fp32-output form: timing at the si_layers shapes, rotating buffers."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis
hip = mrdis.hip
dev = torch.device('cuda:0')


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (N, C, H) in ((32, 32, 256), (32, 64, 128), (32, 128, 64), (128, 128, 32)):
    xs = [torch.randn(N, 4, H, H, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    wt = torch.randn(9, 4, C, device=dev) * 0.1
    wtp = F.pad(wt, (0, 0, 0, 12))
    b = torch.randn(C, device=dev) * 0.1
    y32 = [hip.empty_nhwc(N, C, H, H, dev) for _ in range(4)]
    y16 = [hip.empty_nhwc(N, C, H, H, dev, torch.bfloat16) for _ in range(4)]
    k = [0]

    def f32():
        k[0] += 1
        hip.conv2d_fwd(xs[k[0] % 4], wt, b, 3, 3, 1, 1, out=y32[k[0] % 4])

    def f16():
        k[0] += 1
        hip.conv2d_fwd(xs[k[0] % 4], wtp, b, 3, 3, 1, 1, out=y16[k[0] % 4])
    t32, t16 = timeit(f32), timeit(f16)
    hip.conv2d_fwd(xs[0], wt, b, 3, 3, 1, 1, out=y32[0]); hip.conv2d_fwd(xs[0], wtp, b, 3, 3, 1, 1, out=y16[0])
    err = float((y16[0].float() - y32[0]).abs().max()) / float(y32[0].abs().max())
    print(f'N={N} 4->{C} {H}x{H}: fp32 out {t32:7.1f} us   bf16 out {t16:7.1f} us   max rel diff {err:.2e}')
