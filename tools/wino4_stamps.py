"""Where a wino4_kernel iteration spends its time: in-kernel s_memtime stamps of workgroups 0-3 (ab/libmrdis_abl.so, variant 64; diagnosis only).
Per wave and iteration: top -> last MFMA issued -> end-of-iteration wait done (S waves: vmcnt) -> barrier passed; per block the epilogue.

    python tools/wino4_stamps.py [N Ci Co H W]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

lib = mrdis.hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl.so'))
from tools.wino4_check import images  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
N, ci, co, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 128, 256, 64, 64)
hip.set_option('wino', 2); hip.set_option('wino4', 2)
x = torch.randn(N, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(9, ci, co, device=dev) * 0.05
bias = torch.randn(co, device=dev)
im_f, _ = images(wt, wt.permute(0, 2, 1).contiguous(), dev)
for _ in range(20):
    hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_wino=im_f)
CAP = 4096
buf = torch.zeros(4 * 8 * CAP, dtype=torch.int64, device=dev)
lib.mrdis_debug_wino4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mrdis_debug_wino4_stamps(buf.data_ptr(), CAP)
hip.set_option('debug_mode', 64 | int(os.environ.get('W4_ABL', '0')))
hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_wino=im_f)
torch.cuda.synchronize()
hip.set_option('debug_mode', -1)
st = buf.cpu().numpy().reshape(4, 8, CAP)
nch = ci // 4
print(f'{N}x{ci}->{co} {H}x{W}: {nch} iterations per block; cycles are s_memtime ticks (shader clock)')
for wg in range(2):
    for wave in (0, 1, 4, 5):
        v = st[wg, wave]; v = v[v != 0]
        t = (v >> 4).astype(np.int64); tag = (v & 15).astype(np.int64)
        it_top = t[tag == 1]; it_mf = t[tag == 2]; it_wait = t[tag == 3]; it_bar = t[tag == 4]; ep0 = t[tag == 5]; ep1 = t[tag == 6]
        n = min(len(it_top), len(it_bar))
        body = (it_mf[:n] - it_top[:n]); wait = (it_wait[:n] - it_mf[:n]); bar = (it_bar[:n] - it_wait[:n])
        tot = it_top[1:n] - it_top[:n - 1]
        ne = min(len(ep0), len(ep1))
        epi = ep1[:ne] - ep0[:ne]
        span = (t.max() - t.min())
        print(f'wg {wg} wave {wave} ({"T" if wave < 4 else "S"}): {n} iterations, kernel span {span} cyc; per iteration median: steps {np.median(body):.0f}, '
              f'end wait {np.median(wait):.0f} (p90 {np.percentile(wait, 90):.0f}, max {wait.max()}), barrier {np.median(bar):.0f} (p90 {np.percentile(bar, 90):.0f}), '
              f'top-to-top {np.median(tot):.0f}; epilogues {ne} x {np.median(epi) if ne else 0:.0f} cyc; sums: steps {body.sum()} wait {wait.sum()} barrier {bar.sum()} epilogue {epi.sum()}')
        if wave in (0, 4) and wg == 0:
            k = nch  # the first iteration after the first epilogue
            print('     first block + 2 iterations, (steps, wait, barrier) per iteration: ' + ' '.join(f'({body[i]},{wait[i]},{bar[i]})' for i in range(min(n, k + 3))))
