"""debug: does the shared pool matter?  order of first replays?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis as m
from mrdis import ops
dev = torch.device('cuda:0')
M, B, H, W = 3, 8, 64, 96
def run(graph, order, private_pools=False):
    cfg = dict(m.DEFAULT_CONFIG); cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=16, lambda_adv_s=1.0)
    cfg = m.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = m.build_model(cfg).train()
    base = m.TrainStep(model, cfg)
    step = m.GraphedTrainStep(base, warm=1) if graph else base
    if graph and private_pools:
        orig = step._record
        step._record = lambda x, mm, mi, mh, tg, ds, pool: orig(x, mm, mi, mh, tg, ds, None)
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
            ops.set_forced_pairs(pairs); loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask); ops.set_forced_pairs(None)
        out.append((pair, float(loss), float(base.optimizer.flat_p.double().abs().sum()), float(base.last_grad_norm_sq[0])))
    return out
for order in ([(0, 1), (2, 1), (2, 1), (0, 1), (1, 0)], [(0, 1), (0, 1), (2, 1), (0, 2)]):
    e = run(False, order)
    for pp in (False, True):
        g = run(True, order, pp)
        print('order', order, 'private pools' if pp else 'shared pool', ['same' if a == b else 'DIFF' for a, b in zip(e, g)])
