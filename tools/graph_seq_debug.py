"""debug: eager vs graph over an `it` sequence with an epoch restart (F, F, T) and an optional EvalStep in between"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis as m
dev = torch.device('cuda:0')
M, B, H, W = 3, 8, 64, 96
def run(graph, with_eval):
    cfg = dict(m.DEFAULT_CONFIG); cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=8, lambda_adv_s=1.0)
    cfg = m.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = m.build_model(cfg).train()
    step = m.TrainStep(model, cfg)
    ev = m.EvalStep(model, cfg)
    if graph: step = m.GraphedTrainStep(step)
    torch.manual_seed(100); np.random.seed(100)
    out = []
    seq = [0, 1, 2, 3, 4, 'eval', 0, 1, 2, 3, 4]
    k = 0
    for it in seq:
        x, mask, mask_img = m.synthetic_batch(B, M, H, W, seed=60 + k); k += 1
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        if it == 'eval':
            if with_eval: ev(xd, mask.to(dev), mask_img.to(dev), mask)
            continue
        loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask, it=it)
        o = step.optimizer
        acc = step.acc
        out.append((it, float(loss), float(o.flat_p.double().abs().sum()), float(acc.double().abs().sum()), float(o.m.double().abs().sum()), float(step.optimizer_d_s.m.double().abs().sum())))
    return out
for with_eval in (False, True):
    a, b = run(False, with_eval), run(True, with_eval)
    print('with_eval', with_eval)
    for x, y in zip(a, b):
        print('  same' if x == y else '  DIFF', x, y if x != y else '')
