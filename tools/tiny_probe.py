"""bf16 forward / data gradient of the small-map layers under tile-shape overrides (option debug_mode = 0: the big tiles everywhere, as before the small-map policy)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis
hip = mrdis.hip
dev = torch.device('cuda:0')
B16 = torch.bfloat16
L = [('sp1.gb', 128, 256, 8), ('sp1.out', 128, 128, 8), ('sp2.gb', 128, 256, 16), ('sp2.out', 128, 128, 16), ('up_4', 256, 256, 16),
     ('sp3.gb', 128, 256, 32), ('sp3.out', 128, 128, 32), ('up_3', 512, 128, 32)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, ci, co, hw in L:
    x = torch.randn(32, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    dy = torch.randn(32, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).to(B16)
    wt = torch.randn(9, ci, co, device=dev) * 0.05
    wk = wt.permute(0, 2, 1).contiguous()
    bias = torch.zeros(co, device=dev)
    wf, wb = hip.cast_bf16(wk), hip.cast_bf16(wt)
    row = []
    ref = None
    for mode in (0, -1):
        hip.set_option('debug_mode', mode)
        tf = timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wf))
        td = timeit(lambda: hip.conv2d_bwd_data(dy, wk, (hw, hw), 3, 3, 1, 1, w_bf16=wb))
        y = hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wf)
        if ref is None:
            ref = y
        row.append(f'mode {mode:2d}: fwd {tf:6.1f} dgrad {td:6.1f} {"same" if torch.equal(y, ref) else "DIFF"}')
    hip.set_option('debug_mode', -1)
    print(f'{name:8s} {ci:4d}->{co:4d} {hw:3d}^2 | ' + ' | '.join(row))
