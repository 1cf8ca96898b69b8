"""Where an item of the six-product tap kernel spends its time: s_memtime stamps of workgroups 0-3 (ab/libmrdis_abl_s6t.so; diagnosis only).
Per wave and item (position tile x channel chunk): loop top -> barrier A passed -> images stored (split + LDS writes) -> barrier B passed -> next item's
global loads issued -> MFMA loop done -> epilogue done.      python tools/s6conv_stamps.py [Ci Co k stride H] [fwd|dgrad]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

lib = mrdis.hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl_s6t.so'))
hip = mrdis.hip
dev = torch.device('cuda:0')
Ci, Co, k, st, H = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 64, 4, 2, 128)
what = sys.argv[6] if len(sys.argv) > 6 else 'fwd'
B, pad = 32, 1
x = torch.randn(B, Ci, H, H, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(Co, Ci, k, k) * 0.05
wt = w.permute(2, 3, 1, 0).reshape(-1, Ci, Co).contiguous().to(dev); wk = w.permute(2, 3, 0, 1).reshape(-1, Co, Ci).contiguous().to(dev)
Ho = (H + 2 * pad - k) // st + 1
dy = torch.randn(B, Co, Ho, Ho, device=dev).contiguous(memory_format=torch.channels_last)
bias = torch.randn(Co, device=dev)
imf, imd = hip.s6_filter_image(wt), hip.s6_filter_image(wk)
fn = (lambda: hip.conv2d_fwd(x, wt, bias, k, k, st, pad, w_wino=imf)) if what == 'fwd' else (lambda: hip.conv2d_bwd_data(dy, wk, (H, H), k, k, st, pad, w_wino=imd))
hip.set_option('split6', 10)
for _ in range(10):
    fn()
CAP = 1024
buf = torch.zeros(4 * 4 * CAP, dtype=torch.int64, device=dev)
hip.set_option('debug_bm', buf.data_ptr()); hip.set_option('debug_bn', CAP)
fn()
torch.cuda.synchronize()
hip.set_option('debug_bm', -1); hip.set_option('debug_bn', -1)
st_ = buf.cpu().numpy().reshape(4, 4, CAP)
names = {1: 'top', 2: 'barrier A', 3: 'stored', 4: 'barrier B', 5: 'loads issued', 6: 'MFMA loop', 7: 'epilogue'}
print(f'{what} {Ci}->{Co} k{k} s{st} {H}x{H}: s_memtime ticks (shader clock: 241k ticks = the 115 us of the launch)')
for wg in range(2):
    for wave in (0, 3):
        v = st_[wg, wave]; v = v[v != 0]
        t = (v >> 4).astype(np.int64); tag = (v & 15).astype(np.int64)
        nit = int((tag == 1).sum())
        seg = {}
        for i in range(len(t) - 1):
            seg.setdefault((int(tag[i]), int(tag[i + 1])), []).append(int(t[i + 1] - t[i]))
        total = int(t[-1] - t[0])
        print(f'  wg {wg} wave {wave}: {nit} items, {total} ticks = {total / max(nit, 1):.0f} per item')
        for (a, b), d in sorted(seg.items()):
            print(f'      {names[a]:>12s} -> {names[b]:<12s} n = {len(d):3d}  mean {np.mean(d):7.1f}  min {min(d):5d}  max {max(d):5d}  share {100.0 * sum(d) / total:5.1f} %')
