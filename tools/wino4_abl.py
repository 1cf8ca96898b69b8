"""Timing-only ablations of wino4_kernel (ab/libmrdis_abl.so: the library with mrdis_wino4.hip built -DWINO4_ABLATIONS; option debug_mode
selects the variant; results of the ablated variants are wrong by construction).

    (cd representation-disentanglement_amd/csrc && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DWINO4_ABLATIONS -c -o ../../ab/mrdis_wino4_abl.o mrdis_wino4.hip &&
     hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/libmrdis_abl.so $(ls *.o | grep -v mrdis_wino4.o) ../../ab/mrdis_wino4_abl.o)
    python tools/wino4_abl.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

mrdis.hip.load(os.path.join(ROOT, 'ab', 'libmrdis_abl.so'))
from tools.wino4_check import images, timeit  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
hip.set_option('wino', 2); hip.set_option('wino4', 2)
for (B, ci, co, H, W) in [(32, 128, 256, 64, 64), (32, 64, 128, 128, 128), (32, 32, 64, 256, 256)]:
    x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(9, ci, co, device=dev) * 0.05
    wk = wt.permute(0, 2, 1).contiguous()
    bias = torch.randn(co, device=dev)
    im_f, _ = images(wt, wk, dev)
    out = []
    for abl, name in ((-1, 'full'), (1, 'no V transform'), (8, 'no filter DMA'), (32, 'no raw loads'), (41, 'no V, no global'),
                      (45, 'barriers + operand reads only'), (4, 'no MFMA'), (-1, 'full again')):
        hip.set_option('debug_mode', abl)
        out.append(f'{name} {timeit(lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_wino=im_f)):.1f}')
    hip.set_option('debug_mode', -1)
    print(f'{B}x{ci}->{co} {H}x{W}: ' + ' | '.join(out), flush=True)
