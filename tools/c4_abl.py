"""Timing-only ablations of c4conv_kernel (builds with -DC4_ABL_NOSTORE / _NOMFMA / _NOLOAD, tools/micro/libmrdis_*.so) against the product
library, fp32-out and bf16-out, in one process:  x (32, 4, 256, 256) fp32 -> y (32, 32, 256, 256)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis
from mrdis import hip
dev = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.abspath(__file__))


def bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in hip._SIGS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib


libs = {'product': hip.load()}
for v in ('NOSTORE', 'NOMFMA', 'NOLOAD'):
    p = os.path.join(ROOT, 'micro', f'libmrdis_{v}.so')
    if os.path.exists(p):
        libs[v] = bind(p)
N, H, W = 32, 256, 256
for Co, Ci_w in ((32, 16), (64, 16)):
    x = torch.randn(N, 4, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w4 = torch.randn(9, 4, Co, device=dev) * 0.1
    w16 = torch.nn.functional.pad(w4, (0, 0, 0, 12)).contiguous()
    b = torch.zeros(Co, device=dev)
    y32 = hip.empty_nhwc(N, Co, H, W, dev)
    y16 = hip.empty_nhwc(N, Co, H, W, dev, torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    for name, lib in libs.items():
        r = []
        for tag, w, y, dt in (('f32 out', w4, y32, hip.DT_F32), ('bf16 out', w16, y16, hip.DT_XF32_YBF16)):
            def run():
                rc = lib.mrdis_conv2d_fwd(x.data_ptr(), 4, w.data_ptr(), None, b.data_ptr(), y.data_ptr(), Co, N, H, W, 4, Co, 3, 3, 1, 1, 0, dt, None, 0, st)
                assert rc == 0, rc
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run()
            e1.record(); torch.cuda.synchronize()
            r.append(f'{tag} {e0.elapsed_time(e1) * 1e3 / 30:6.1f} us')
        print(f'4 -> {Co}: {name:8s} ' + ' | '.join(r))
