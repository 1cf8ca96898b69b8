"""A/B two builds of libmrdis_hip.so in ONE process on ONE device (devices and boxes differ by several
percent, so timings from separate gpurun calls do not compare):
    python tools/ab_lib.py libA.so libB.so [--shape N Ci H W Co] [--iters 50] [--rounds 5]
Times mrdis_conv2d_fwd (3x3 s1 p1) alternately through both libraries."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402


def bind(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (res, args) in hip._SIGS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('libs', nargs='+')
    ap.add_argument('--shape', type=int, nargs=5, default=[32, 4, 240, 240, 32])
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--op', default='fwd', choices=['fwd', 'wgrad'])
    ap.add_argument('--k', type=int, default=3)
    ap.add_argument('--stride', type=int, default=1)
    a = ap.parse_args()
    N, Ci, H, W, Co = a.shape
    dev = torch.device('cuda:0')
    x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(a.k * a.k, Ci, Co, device=dev) * 0.1
    b = torch.randn(Co, device=dev)
    y = hip.empty_nhwc(N, Co, H, W, dev)
    libs = [bind(p) for p in a.libs]
    st = torch.cuda.current_stream().cuda_stream

    k, sd = a.k, a.stride
    pad = 0 if k == 1 else 1
    Ho, Wo = (H + 2 * pad - k) // sd + 1, (W + 2 * pad - k) // sd + 1
    if a.op == 'wgrad':
        dy = torch.randn(N, Co, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
        dw = torch.empty(k * k, Ci, Co, device=dev); db = torch.empty(Co, device=dev)
        nb = max(int(l.mrdis_conv2d_bwd_weight_workspace(N, H, W, Ci, Co, k, k, sd, pad)) for l in libs)
        ws = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
        y = dw

    def run(lib):
        if a.op == 'wgrad':
            rc = lib.mrdis_conv2d_bwd_weight(x.data_ptr(), Ci, dy.data_ptr(), Co, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb + 256,
                                             N, H, W, Ci, Co, k, k, sd, pad, 0, st)
        else:
            rc = lib.mrdis_conv2d_fwd(x.data_ptr(), Ci, w.data_ptr(), b.data_ptr(), y.data_ptr(), Co, N, H, W, Ci, Co, k, k, sd, pad, 0, st)
        assert rc == 0, rc
    outs = []
    for lib in libs:
        run(lib); torch.cuda.synchronize(); outs.append(y.clone())
    for o in outs[1:]:
        print('max |diff| vs first lib:', float((o - outs[0]).abs().max()))
    best = [1e9] * len(libs)
    for r in range(a.rounds):
        line = []
        for i, lib in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(lib); torch.cuda.synchronize()
            e0.record()
            for _ in range(a.iters):
                run(lib)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            best[i] = min(best[i], us); line.append(f'{us:8.2f}')
        print('round', r, ' '.join(line), flush=True)
    print('best us:', ' '.join(f'{b_:8.2f}' for b_ in best))


if __name__ == '__main__':
    main()
