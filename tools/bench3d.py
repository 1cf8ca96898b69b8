"""3-D path timing (BASELINE.json configs[4]: synthetic 4x128x128x128 volumes, batch 4, 1 GPU): one NVNet3D training
step (forward, nvnet_loss, backward, Adam on the flat arena) and, with --layers, every distinct Conv3d geometry of the
net (forward / data gradient / weight gradient, TFLOP/s against the 157 TFLOP/s fp32 MFMA peak).

    python tools/bench3d.py [--size 128] [--batch 4] [--channels 16] [--steps 5] [--layers]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def layer_table(B, S, c, reps):
    dev = torch.device('cuda:0')
    geoms = [('conv1a', 4, c, S, 1)]
    for lvl, mult in enumerate((1, 2, 4, 8)):
        s = S >> lvl
        geoms.append((f'block{lvl + 1} {c * mult}->{c * mult}', c * mult, c * mult, s, 1))
        if lvl < 3:
            geoms.append((f'ds{lvl + 1}', c * mult, c * mult * 2, s, 2))
    for lvl, mult in enumerate((8, 4, 2)):
        geoms.append((f'vconv{3 - lvl} {c * mult}->{c * mult // 2}', c * mult, c * mult // 2, S >> (3 - lvl), 1))
    geoms.append(('hidden_conv', c * 8, c * 4, S >> 3, 1))
    rows = []
    for name, ci, co, s, st in geoms:
        x = torch.randn(B, s, s, s, ci, device=dev).permute(0, 4, 1, 2, 3)
        w = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.1
        one = torch.ones(1, device=dev)
        w_tck, w_tkc = mrdis.hip.mix_experts_fwd(w.reshape(1, co, ci, 27, 1), one)
        bias = torch.zeros(co, device=dev)
        y = mrdis.hip.conv3d_fwd(x, w_tck, bias, 3, st, 1)
        dy = torch.randn_like(y)
        so = y.shape[2]
        flop = 2.0 * B * so ** 3 * 27 * ci * co
        t_f = timeit(lambda: mrdis.hip.conv3d_fwd(x, w_tck, bias, 3, st, 1), reps)
        t_d = timeit(lambda: mrdis.hip.conv3d_bwd_data(dy, w_tkc, tuple(x.shape), 3, st, 1), reps)
        t_w = timeit(lambda: mrdis.hip.conv3d_bwd_weight(x, dy, 3, st, 1, True), reps)
        mb = 4.0 * (x.numel() + y.numel()) / 1e6
        # roofline leg (round 6): direct-convolution FLOPs / time against the fp32 MFMA peak (157.3 TF/s; the hybrid Winograd and the six-product kernels execute
        # fewer / cheaper multiplies, so their fraction can exceed what an fp32 MFMA kernel could reach) and algorithmic bytes (x + y, fp32) / time against 8 TB/s
        frac = lambda t: (round(flop / t / 1e6 / 157.3, 3), round(mb / t * 1e3 / 8000.0, 3))
        rows.append(dict(layer=name, ci=ci, co=co, size=s, stride=st, gflop=round(flop / 1e9, 2), mb=round(mb, 1),
                         fwd_us=round(t_f, 1), dgrad_us=round(t_d, 1), wgrad_us=round(t_w, 1),
                         fwd_tf=round(flop / t_f / 1e6, 1), dgrad_tf=round(flop / t_d / 1e6, 1), wgrad_tf=round(flop / t_w / 1e6, 1),
                         fwd_gbs=round(mb / t_f * 1e3, 0),
                         fwd_frac_mfma_hbm=frac(t_f), dgrad_frac_mfma_hbm=frac(t_d), wgrad_frac_mfma_hbm=frac(t_w),
                         bound='mfma' if flop / (mb * 1e6) > 157.3e12 / 8e12 else 'hbm'))
        print(json.dumps(rows[-1]), flush=True)
        del x, y, dy
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--channels', type=int, default=16)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--layers', action='store_true')
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    assert torch.cuda.is_available(), 'needs an MI355X (no CPU fallback)'
    dev = torch.device('cuda:0')
    mrdis.hip.load()
    if a.layers:
        layer_table(a.batch, a.size, a.channels, a.reps)
        return
    S, B = a.size, a.batch
    torch.manual_seed(10)
    model = mrdis.NVNet3D((S, S, S), 4, 3, a.channels, p=0.2).to(dev).train()
    opt = mrdis.ArenaAdam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, S, S, S, 4, generator=g).to(dev).permute(0, 4, 1, 2, 3)
    t = (torch.rand(B, S, S, S, 3, generator=g) > 0.7).float().to(dev).permute(0, 4, 1, 2, 3)

    def step():
        out = model(x)
        loss, _ = mrdis.nvnet_loss(*out, x, t)
        loss.backward()
        opt.step(fused_clip=True)
        opt.zero_grad()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({'metric': 'NVNet3D train step', 'ms_per_step': round(dt * 1e3, 2), 'volumes_per_s': round(B / dt, 2),
                      'config': {'workload': f'{B}x4x{S}^3 fp32, init_channels {a.channels}, dropout 0.2'},
                      'loss': float(loss), 'peak_mem_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))


if __name__ == '__main__':
    main()
