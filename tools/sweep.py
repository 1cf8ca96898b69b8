"""Sweep tapconv block configurations (cout tile BN x channel chunk KC x staging mode x positions-per-workgroup policy)
per layer shape, fwd and dgrad, through the library's MRDIS_DEBUG_{BN,KC,MODE,BM} knobs."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis
from tools.layer_bench import layers, timeit
hip = mrdis.hip
dev = torch.device('cuda:0')
B, H, W = 32, 256, 256
CFG = [(bn, kc, pf, bm) for bn in (32, 64) for kc in (8, 16) for pf in (0, 1) for bm in (0, 2)]
seen = set()
for name, ci, co, k, s, p, hi, wi, calls, dcalls in layers(B, H, W):
    if ci < 16 or co < 16 or (ci, co, k, s, hi) in seen:
        continue
    seen.add((ci, co, k, s, hi))
    x = torch.randn(B, ci, hi, wi, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(k * k, ci, co, device=dev) * 0.05
    wk = wt.permute(0, 2, 1).contiguous()
    ho, wo = hip.conv_out_hw(hi, wi, k, k, s, p)
    dy = torch.randn(B, co, ho, wo, device=dev).contiguous(memory_format=torch.channels_last)
    for kind, fn in (('fwd', lambda: hip.conv2d_fwd(x, wt, None, k, k, s, p)), ('dgrad', lambda: hip.conv2d_bwd_data(dy, wk, (hi, wi), k, k, s, p))):
        res = []
        for bn, kc, pf, bm in CFG:
            cout = co if kind == 'fwd' else ci
            if bn > 32 and bn // 2 >= cout:
                continue
            hip.set_option('debug_bn', bn); hip.set_option('debug_kc', kc); hip.set_option('debug_mode', pf); hip.set_option('debug_bm', bm)
            try:
                res.append((timeit(fn, 3), bn, kc, pf, bm))
            except Exception as e:
                pass
        for v in ('debug_bn', 'debug_kc', 'debug_mode', 'debug_bm'):
            hip.set_option(v, -1)
        dflt = timeit(fn, 3)
        res.sort()
        flop = 2.0 * k * k * ci * co * B * ho * wo
        print(f'{name:16s} {kind:5s} ci={ci:3d} co={co:3d} k{k}s{s} {hi:3d}: default {dflt:7.1f}us | best ' +
              '  '.join(f'BN{bn}/KC{kc}/mode{pf}/bm{bm}={t:6.1f}({flop / t / 1e6:5.1f}TF)' for t, bn, kc, pf, bm in res[:3]), flush=True)
