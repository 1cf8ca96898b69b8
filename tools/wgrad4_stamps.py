"""In-kernel stamps of wino4_wgrad_kernel (ab/libmrdis_abl.so): per wave and iteration: top -> last MFMA issued -> barrier passed.   python tools/wgrad4_stamps.py [N Ci Co H W]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

lib = mrdis.hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl.so'))
hip = mrdis.hip
dev = torch.device('cuda:0')
N, ci, co, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 128, 256, 64, 64)
hip.set_option('wino', 2); hip.set_option('wino4', 2)
x = torch.randn(N, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn(N, co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(10):
    hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1)
CAP = 2048
buf = torch.zeros(4 * 8 * CAP, dtype=torch.int64, device=dev)
lib.mrdis_debug_wino4w_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mrdis_debug_wino4w_stamps(buf.data_ptr(), CAP)
hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1)
torch.cuda.synchronize()
lib.mrdis_debug_wino4w_stamps(None, 0)
st = buf.cpu().numpy().reshape(4, 8, CAP)
for wg in range(1):
    for wave in (0, 1, 4, 5):
        v = st[wg, wave]; v = v[v != 0]
        t = (v >> 4).astype(np.int64); tag = (v & 15).astype(np.int64)
        a, b, c, d = t[tag == 1], t[tag == 2], t[tag == 3], t[tag == 4]
        n = min(len(a), len(d))
        print(f'wg {wg} wave {wave} ({"V" if wave < 4 else "Z"}): {n} iterations; median steps {np.median(b[:n] - a[:n]):.0f}, barrier {np.median(d[:n] - c[:n]):.0f} (p90 {np.percentile(d[:n] - c[:n], 90):.0f}), '
              f'top-to-top {np.median(a[1:n] - a[:n - 1]):.0f}; first ten steps: {(b[:10] - a[:10]).tolist()}')
