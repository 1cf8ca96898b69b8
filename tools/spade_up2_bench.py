"""SPADE backward with the x2 resize's adjoint inside (mrdis_instnorm_spade_bwd_up2, one-pass form with xlo) at the step's shapes, fp32 and bf16:
   this build (512 threads) | this build with debug_mode 2002 (256 threads) | an older build given as argv[1] | the unfused pair (SPADE backward + resize adjoint).
   python tools/spade_up2_bench.py [old_lib.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402
from tools.ab_lib import bind  # noqa: E402

dev = torch.device('cuda:0')
hip.load()
old = bind(sys.argv[1]) if len(sys.argv) > 1 else None


def timed(fn, reps=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


for dt in (torch.float32, torch.bfloat16):
    for (N, C, Hi) in [(128, 32, 128), (128, 64, 64), (32, 32, 128), (32, 128, 32)]:
        H = W = 2 * Hi
        torch.manual_seed(0)
        x = cl(torch.randn(N, C, Hi, Hi, device=dev)).to(dt)
        z = hip.bilinear_fwd(x, (H, W), False)
        gb = cl(torch.randn(N, 2 * C, H, W, device=dev)).to(dt)
        gamma = gb[:, :C]
        dout = cl(torch.randn(N, C, H, W, device=dev)).to(dt)
        zf = z.float()
        mean = zf.mean((2, 3)).reshape(-1).contiguous(); rstd = (zf.var((2, 3), unbiased=False) + 1e-5).rsqrt().reshape(-1).contiguous()
        del zf
        res = {}
        hip.set_option('debug_mode', -1)
        res['new512'] = timed(lambda: hip.instnorm_spade_bwd(dout, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x))
        dx_new, dgb_new = hip.instnorm_spade_bwd(dout, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x)
        hip.set_option('debug_mode', 2002)
        res['new256'] = timed(lambda: hip.instnorm_spade_bwd(dout, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x))
        hip.set_option('debug_mode', -1)

        def unfused():
            dz, dgb = hip.instnorm_spade_bwd(dout, z, gamma, mean, rstd, fused_gb=True)
            return hip.bilinear_bwd(dz, (Hi, Hi), False), dgb
        res['unfused'] = timed(unfused)
        dx_ref, dgb_ref = unfused()
        err = float((dx_new.float() - dx_ref.float()).abs().max()) / float(dx_ref.float().abs().max())
        errg = float((dgb_new[:, :C].float() - dgb_ref[:, :C].float()).abs().max()) / float(dgb_ref[:, :C].float().abs().max())
        if old is not None:
            d = hip.DT_F32 if dt is torch.float32 else hip.DT_BF16
            nb = old.mrdis_instnorm_spade_bwd_up2_workspace(N, Hi, Hi, C, d)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            dx = hip.empty_nhwc(N, C, Hi, Hi, dev, dt); dgb = hip.empty_nhwc(N, 2 * C, H, W, dev, dt)
            es = dgb.element_size()
            st = torch.cuda.current_stream().cuda_stream

            def run_old():
                rc = old.mrdis_instnorm_spade_bwd_up2(dout.data_ptr(), C, None, 0, gamma.data_ptr(), 2 * C, mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), C,
                                                      dgb.data_ptr(), 2 * C, dgb.data_ptr() + es * C, 2 * C, ws.data_ptr(), nb, N, Hi, Hi, C, x.data_ptr(), C, d, st)
                assert rc == 0, rc
            res['old'] = timed(run_old)
        el = N * H * W * C
        byt = el * dout.element_size() * 3          # dout + gamma read, dgamma written (+ dbeta copy when not in place: counted by neither)
        print(f'{str(dt):15s} N={N:3d} C={C:3d} {H}x{W}: ' + ' | '.join(f'{k} {v:7.1f} us' for k, v in res.items()) +
              f' | new512 = {byt / res["new512"] / 1e6:.2f} TB/s of the 3 full-resolution streams | dx err {err:.1e} dgamma err {errg:.1e}', flush=True)
