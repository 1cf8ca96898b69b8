"""debug: every adv_s variant graph replayed in a chosen order vs the eager step with the same forced pairs"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis as m
from mrdis import ops
dev = torch.device('cuda:0')
M, B, H, W = 3, 8, 64, 96
order = [(0, 1), (0, 1), (2, 1), (2, 1), (1, 0), (0, 2), (1, 2), (2, 0), (0, 1), (2, 1)]
def run(graph, bs):
    cfg = dict(m.DEFAULT_CONFIG); cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=bs, lambda_adv_s=1.0)
    cfg = m.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = m.build_model(cfg).train()
    base = m.TrainStep(model, cfg)
    step = m.GraphedTrainStep(base, warm=1) if graph else base
    torch.manual_seed(100); np.random.seed(100)
    out = []
    for k, pair in enumerate(order):
        x, mask, mask_img = m.synthetic_batch(B, M, H, W, seed=60 + k)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        pairs = {'sim_s': (1, 2), 'adv_s': pair}
        if graph:
            step._predraw = lambda p=pairs: dict(p)
            loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask)
        else:
            ops.set_forced_pairs(pairs)
            loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask)
            ops.set_forced_pairs(None)
        o = base.optimizer
        out.append((pair, float(loss), float(o.flat_p.double().abs().sum()), None if base.acc is None else float(base.acc.double().abs().sum()), float(base.last_grad_norm_sq[0]), float(base.last_grad_norm_sq[1])))
    return out
for bs in (16, 8):
    a, b = run(False, bs), run(True, bs)
    print('batch_size', bs)
    for x, y in zip(a, b):
        print('  same' if x == y else '  DIFF', x, y if x != y else '')
