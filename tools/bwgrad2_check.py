"""bf16-storage weight gradient: the pipelined kernel (bwgrad2_kernel, option wino_pipe = 1) against bwgrad_kernel (wino_pipe = 0):
bit-identical results expected; time per call."""
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
    shapes = [(2, 32, 64, 3, 50, 37), (3, 64, 40, 3, 50, 33), (4, 96, 32, 1, 40, 70), (32, 32, 64, 3, 256, 256), (32, 64, 128, 3, 128, 128), (32, 128, 256, 3, 64, 64),
              (32, 128, 64, 3, 64, 64), (32, 128, 256, 3, 32, 32), (8, 512, 128, 3, 32, 32), (32, 32, 16, 3, 256, 256), (32, 64, 32, 3, 128, 128),
              (32, 128, 32, 3, 128, 128), (32, 64, 16, 3, 256, 256), (32, 128, 128, 3, 16, 16)]
    if len(sys.argv) > 1 and sys.argv[1] == 'small':
        shapes = shapes[:3]
    for (B, ci, co, k, H, W) in shapes:
        pad = k // 2
        x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
        dy = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
        res = {}
        for pipe in (0, 1):
            hip.set_option('wino_pipe', pipe)
            dw, db = hip.conv2d_bwd_weight(x, dy, k, k, 1, pad, need_bias=True)
            t = timeit(lambda: hip.conv2d_bwd_weight(x, dy, k, k, 1, pad, need_bias=True))
            res[pipe] = (dw, db, t)
        hip.set_option('wino_pipe', 1)
        same = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        err = ((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()).item()
        print(f'{B}x{ci}->{co} k{k} {H}x{W}: bwgrad {res[0][2]:7.1f} us | pipelined {res[1][2]:7.1f} us | bit-identical {same} (max rel diff {err:.1e})', flush=True)


if __name__ == '__main__':
    main()
