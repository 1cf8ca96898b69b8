import os, sys, torch
sys.path.insert(0, '/root/repo')
import mrdis
from mrdis import hip
from tools.wino2_check import timeit
dev = torch.device('cuda:0'); hip.load()
for (N, Co, H, W) in [(32, 32, 256, 256), (32, 64, 128, 128), (32, 128, 64, 64)]:
    x = torch.randn(N, 4, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16)
    fn = lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, may_decline=True)
    r = {}
    for mode in (3011, -1):
        hip.set_option('debug_mode', mode); r[mode] = timeit(fn, iters=20)
    hip.set_option('debug_mode', -1)
    print(f'4->{Co} {H}x{W}: widening fp32 form {r[3011]:.1f} us | bf16 pipe {r[-1]:.1f}', flush=True)
