"""bf16-storage 3x3 convolution: the pipelined kernel (bconv3_kernel, option wino_pipe = 1) against bconv_kernel (wino_pipe = 0):
bit-identical results expected (same products, same accumulation order); time per call forward / data gradient."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from tools.wino2_check import timeit  # noqa: E402

hip = mrdis.hip


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    B16 = torch.bfloat16
    shapes = [(2, 32, 64, 23, 37), (3, 64, 40, 50, 33), (1, 96, 16, 9, 70), (32, 32, 64, 256, 256), (32, 64, 128, 128, 128), (32, 128, 256, 64, 64),
              (32, 128, 64, 64, 64), (32, 128, 256, 32, 32), (8, 512, 128, 32, 32), (32, 32, 16, 256, 256), (32, 64, 32, 128, 128), (32, 128, 32, 128, 128),
              (32, 64, 16, 256, 256)]
    if len(sys.argv) > 1 and sys.argv[1] == 'small':
        shapes = shapes[:3]
    for (B, ci, co, H, W) in shapes:
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
        wt = torch.randn(9, ci, co, device=dev) * 0.05
        wk = wt.permute(0, 2, 1).contiguous()
        bias = torch.randn(co, device=dev)
        dy = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
        wb_f, wb_b = hip.cast_bf16(wk), hip.cast_bf16(wt)
        res = {}
        for pipe in (0, 1):
            hip.set_option('wino_pipe', pipe)
            y = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True, w_bf16=wb_f)
            g = hip.conv2d_bwd_data(dy, wk, (H, W), 3, 3, 1, 1, w_bf16=wb_b) if co % 32 == 0 else y
            tf = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, lrelu=True, w_bf16=wb_f))
            td = timeit(lambda: hip.conv2d_bwd_data(dy, wk, (H, W), 3, 3, 1, 1, w_bf16=wb_b)) if co % 32 == 0 else 0.0
            res[pipe] = (y, g, tf, td)
        hip.set_option('wino_pipe', 1)
        same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        err = (res[0][0].float() - res[1][0].float()).abs().max().item()
        print(f'{B}x{ci}->{co} {H}x{W}: bconv fwd {res[0][2]:7.1f} dgrad {res[0][3]:7.1f} | pipelined fwd {res[1][2]:7.1f} dgrad {res[1][3]:7.1f} | '
              f'bit-identical {same} (max abs diff {err:.2e})', flush=True)


if __name__ == '__main__':
    main()
