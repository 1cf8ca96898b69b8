"""Where a bconv4_kernel item goes: in-kernel s_memtime stamps of workgroups 0-3 (ab/libmrdis_abl_bf16.so, variant 64; diagnosis only).
Per wave and item: top -> last MFMA issued -> own DMA landed (vmcnt) -> barrier passed; per unit the epilogue.   python tools/bconv4_stamps.py [N Ci Co H W] [extra ABL bits]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

lib = mrdis.hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl_bf16.so'))
hip = mrdis.hip
dev = torch.device('cuda:0')
N, ci, co, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (32, 128, 256, 64, 64)
extra = int(sys.argv[6]) if len(sys.argv) > 6 else 0
x = torch.randn(N, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).to(torch.bfloat16)
wt = torch.randn(9, ci, co, device=dev) * 0.05
wb = hip.cast_bf16(wt.permute(0, 2, 1).contiguous())
bias = torch.randn(co, device=dev)
for _ in range(30):
    hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb)
CAP = 4096
buf = torch.zeros(4 * 8 * CAP, dtype=torch.int64, device=dev)
lib.mrdis_debug_bconv4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mrdis_debug_bconv4_stamps(buf.data_ptr(), CAP)
hip.set_option('debug_mode', 64 | extra)
hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb)
torch.cuda.synchronize()
hip.set_option('debug_mode', -1)
st = buf.cpu().numpy().reshape(4, 8, CAP)
nch = ci // 32
print(f'{N}x{ci}->{co} {H}x{W} (ABL {64 | extra}): {nch} items per unit; cycles are s_memtime ticks (shader clock); ideal matrix-pipe time per item: 72 MFMAs x 32 x 2 waves = 4608')
for wg in range(2):
    for wave in (0, 3, 4, 7):
        v = st[wg, wave]; v = v[v != 0]
        t = (v >> 4).astype(np.int64); tag = (v & 15).astype(np.int64)
        it_top = t[tag == 1]; it_mf = t[tag == 2]; it_wait = t[tag == 3]; it_bar = t[tag == 4]; ep0 = t[tag == 5]; ep1 = t[tag == 6]
        n = min(len(it_top), len(it_bar))
        body = it_mf[:n] - it_top[:n]; wait = it_wait[:n] - it_mf[:n]; bar = it_bar[:n] - it_wait[:n]
        tot = it_top[1:n] - it_top[:n - 1]
        ne = min(len(ep0), len(ep1)); epi = ep1[:ne] - ep0[:ne]
        print(f'wg {wg} wave {wave}: {n} items, kernel span {t.max() - t.min()} cyc; per item median: steps {np.median(body):.0f}, DMA wait {np.median(wait):.0f} (p90 {np.percentile(wait, 90):.0f}, '
              f'max {wait.max()}), barrier {np.median(bar):.0f} (p90 {np.percentile(bar, 90):.0f}), top-to-top {np.median(tot):.0f}; epilogues {ne} x {np.median(epi) if ne else 0:.0f} cyc; '
              f'sums: steps {body.sum()} wait {wait.sum()} barrier {bar.sum()} epilogue {epi.sum()}')
        if wave == 0 and wg == 0:
            print('     first items (steps, wait, barrier): ' + ' '.join(f'({body[i]},{wait[i]},{bar[i]})' for i in range(min(n, 10))))
