"""Clock ramp probe: the same convolution timed after different amounts of warm-up (why roofline_step warms every entry point up for >= 20 ms)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import mrdis
hip = mrdis.hip
dev = torch.device('cuda:0')
B, ci, co, h, w = 32, 32, 64, 256, 256
def timeit(fn, n, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(9, ci, co, device=dev) * 0.05
bias = torch.zeros(co, device=dev)
yo = hip.empty_nhwc(B, co, h, w, dev)
print('cold, out=  2 warm 6 it :', timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, out=yo), 6, 2))
print('again                   :', timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, out=yo), 6, 2))
print('50 warm 20 it           :', timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, out=yo), 20, 50))
print('fresh outputs 1 warm 5  :', timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1), 5, 1))
dy = torch.randn(B, co, h, w, device=dev).contiguous(memory_format=torch.channels_last)
dxo = hip.empty_nhwc(B, ci, h, w, dev)
print('with dy, dxo allocated  :', timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, out=yo), 6, 2))
x2 = torch.randn(B, ci, h, w, device=dev).to(torch.float32).contiguous(memory_format=torch.channels_last)
print('other x                 :', timeit(lambda: hip.conv2d_fwd(x2, wt, bias, 3, 3, 1, 1, out=yo), 6, 2))
