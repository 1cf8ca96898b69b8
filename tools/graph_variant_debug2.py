"""debug: which parameters' gradients are garbage when a later-recorded adv_s variant replays"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis as m
from mrdis import ops
dev = torch.device('cuda:0')
M = int(os.environ.get('DBG_M', 3)); B, H, W = 8, 64, 96
order = [(0, 1), (0, 1), (2, 1)]
def run(graph):
    cfg = dict(m.DEFAULT_CONFIG); cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=16, lambda_adv_s=1.0)
    cfg = m.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = m.build_model(cfg).train()
    base = m.TrainStep(model, cfg)
    dbg = {}
    orig = base._apply
    def _apply(scale, do_step):
        dbg[ops.forced_pair('adv_s')] = (base.optimizer._g_full.clone(), base.optimizer_d_s._g_full.clone())
        orig(scale, do_step)
    base._apply = _apply
    step = m.GraphedTrainStep(base, warm=1) if graph else base
    torch.manual_seed(100); np.random.seed(100)
    for k, pair in enumerate(order):
        x, mask, mask_img = m.synthetic_batch(B, M, H, W, seed=60 + k)
        xd = x.to(dev).contiguous(memory_format=torch.channels_last)
        pairs = {'sim_s': (1, 2), 'adv_s': pair}
        if graph:
            step._predraw = lambda p=pairs: dict(p)
            step(xd, mask.to(dev), mask_img.to(dev), mask)
        else:
            ops.set_forced_pairs(pairs); step(xd, mask.to(dev), mask_img.to(dev), mask); ops.set_forced_pairs(None)
    torch.cuda.synchronize()
    g, gd = dbg[order[-1]]
    return base, g.clone(), gd.clone()
be, ge, gde = run(False)
bg, gg, gdg = run(True)
opt = be.optimizer
names = {id(p): n for n, p in be.model.named_parameters()}
bad = []
for p, o in zip(opt.used, opt.offsets):
    k = p.numel()
    a, b = ge[o:o + k], gg[o:o + k]
    if not torch.equal(a, b):
        bad.append((names[id(p)], k, float((a - b).abs().max()), bool(torch.isfinite(b).all())))
print('M', M, 'generator-loss gradient: tensors that differ:', len(bad), 'of', len(opt.used))
for r in bad[:40]: print('  ', r)
bad2 = []
for p, o in zip(opt.used, opt.offsets):
    k = p.numel()
    a, b = gde[o:o + k], gdg[o:o + k]
    if not torch.equal(a, b): bad2.append((names[id(p)], k, float((a - b).abs().max()), bool(torch.isfinite(b).all())))
print('discriminator-loss gradient: tensors that differ:', len(bad2))
for r in bad2[:20]: print('  ', r)
