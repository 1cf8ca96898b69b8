"""bf16 layers with 16 input channels (padded to 32 inside the kernels): timing of the padded path at the bench shapes."""
import sys; sys.path.insert(0, '/root/repo')
import torch, mrdis
hip = mrdis.hip; dev = torch.device('cuda:0'); B16 = torch.bfloat16
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
B = 32
for name, ci, co, k, hw in [('si16->32', 16, 32, 3, 256), ('out64->16', 64, 16, 3, 256), ('head16->16', 16, 16, 1, 256), ('si16->64', 16, 64, 3, 128)]:
    p = (k - 1) // 2
    x = torch.randn(B, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    wt = torch.randn(k * k, ci, co, device=dev) * 0.05; wk = wt.permute(0, 2, 1).contiguous(); bias = torch.zeros(co, device=dev)
    dy = torch.randn(B, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    wbf, wbb = hip.cast_bf16(wk), hip.cast_bf16(wt)
    tf = timeit(lambda: hip.conv2d_fwd(x, wt, bias, k, k, 1, p, w_bf16=wbf))
    td = timeit(lambda: hip.conv2d_bwd_data(dy, wk, (hw, hw), k, k, 1, p, w_bf16=wbb))
    tw = timeit(lambda: hip.conv2d_bwd_weight(x, dy, k, k, 1, p))
    x4 = torch.randn(B, 4, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    tc = timeit(lambda: hip.cast_view(x4, B16, 16)); ts = timeit(lambda: hip.cast_view(dy, torch.float32, 7))
    print(f'{name}: fwd {tf:.1f} dgrad {td:.1f} wgrad {tw:.1f} us | pad-cast 4->16 {tc:.1f} us, slice-cast ->7 fp32 {ts:.1f} us', flush=True)
