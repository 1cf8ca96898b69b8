"""Which loss term makes a graph replay differ from the eager step: fixed sim_s / adv_s pairs, four iterations, one loss weight zeroed at a time.
(Round 6: located the hipMemsetAsync node of max_pool's backward, which did not run again on later replays -- see tests/test_gpu_graph.py.)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis as m
from mrdis import ops
dev = torch.device('cuda:0')
B, H, W = 8, 64, 96
def run(graph, M, over):
    cfg = dict(m.DEFAULT_CONFIG); cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=16, lambda_adv_s=1.0)
    cfg.update(over)
    cfg = m.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = m.build_model(cfg).train()
    base = m.TrainStep(model, cfg)
    step = m.GraphedTrainStep(base, warm=1) if graph else base
    torch.manual_seed(100); np.random.seed(100)
    out = []
    for k in range(4):
        x, mask, mask_img = m.synthetic_batch(B, M, H, W, seed=60 + k)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        pairs = {'sim_s': (1, 2), 'adv_s': (0, 1)}
        if graph:
            step._predraw = lambda p=pairs: dict(p)
            loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask)
        else:
            ops.set_forced_pairs(pairs); loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask); ops.set_forced_pairs(None)
        out.append((float(loss), float(base.optimizer.flat_p.double().abs().sum()), float(base.last_grad_norm_sq[0])))
    return out
for M in (3, 4):
    for name, over in (('all', {}), ('no sim_s', {'lambda_sim_s': 0.0}), ('no adv', {'lambda_adv_s': 0.0}), ('no sim_z', {'lambda_sim_z': 0.0}),
                       ('no latent_z', {'lambda_latent_z': 0.0}), ('no mix', {'lambda_recon_x_mix': 0.0})):
        e, g = run(False, M, over), run(True, M, over)
        print('M', M, name, ['same' if a == b else 'DIFF' for a, b in zip(e, g)], [round(x[2], 3) for x in g])
