"""Where a stage of wino4r_kernel goes: in-kernel s_memtime stamps of workgroups 0-3 (ab/libmrdis_abl.so; diagnosis only).
Per wave and stage: loop top -> copies landed + barrier passed -> chunk(s) done; per block the epilogue.

    python tools/wino4r_stamps.py [N Ci Co H W [mode]]        mode = option wino4r (2: 64-tile form, 3: channel-split pairs)
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
a = [int(v) for v in sys.argv[1:]]
N, ci, co, H, W = a[:5] if len(a) >= 5 else (32, 64, 32, 128, 128)
mode = a[5] if len(a) > 5 else 2
hip.set_option('wino', 2); hip.set_option('wino4', 2); hip.set_option('wino4r', mode)
x = torch.randn(N, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(9, ci, co, device=dev) * 0.05
bias = torch.randn(co, device=dev)
im_f, _ = images(wt, wt.permute(0, 2, 1).contiguous(), dev)
for _ in range(20):
    hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_wino=im_f)
CAP = 4096
buf = torch.zeros(4 * 8 * CAP, dtype=torch.int64, device=dev)
lib.mrdis_debug_wino4r_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mrdis_debug_wino4r_stamps(buf.data_ptr(), CAP)
hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_wino=im_f)
torch.cuda.synchronize()
lib.mrdis_debug_wino4r_stamps(None, 0)
st = buf.cpu().numpy().reshape(4, 8, CAP)
nst = ci // 8
print(f'{N}x{ci}->{co} {H}x{W} wino4r={mode}: {nst} stages per block; cycles are s_memtime ticks (shader clock)')
for wg in range(2):
    for wave in (0, 3, 4, 7):
        v = st[wg, wave]; v = v[v != 0]
        t = (v >> 4).astype(np.int64); tag = (v & 15).astype(np.int64)
        top = t[tag == 1]; bar = t[tag == 2]; done = t[tag == 3]; epi = t[tag == 4]; iss = t[tag == 5]; xf = t[tag == 6]
        n = min(len(top), len(bar), len(done))
        if len(iss) < n: iss = bar
        wait = bar[:n] - top[:n]; comp = done[:n] - bar[:n]; issue = iss[:n] - bar[:n]
        tot = top[1:n] - top[:n - 1]
        per = len(xf) // max(n, 1)                      # chunks per stage for this wave
        xf0 = xf[::per][:n] - iss[:n] if per else np.zeros(1)            # copies issued -> first chunk's patch read + column transform + first row done
        mf = done[:n] - xf[per - 1::per][:n] if per else np.zeros(1)     # last chunk's row steps (36 MFMAs)
        ed = done[nst - 1::nst][:len(epi)]
        ep = epi[:len(ed)] - ed
        print(f'wg {wg} wave {wave}: {n} stages, span {t.max() - t.min()}; per stage median: wait+barrier {np.median(wait):.0f} (p90 {np.percentile(wait, 90):.0f}), '
              f'copy issue {np.median(issue):.0f}, patch+columns {np.median(xf0):.0f}, row steps {np.median(mf):.0f}, compute {np.median(comp):.0f}, top-to-top {np.median(tot):.0f}; '
              f'epilogues {len(ep)} x {np.median(ep) if len(ep) else 0:.0f}')
        if wave == 0 and wg == 0:
            print('     first stages (wait, issue, compute): ' + ' '.join(f'({wait[i]},{issue[i]},{comp[i]})' for i in range(min(n, nst + 3))))
