"""A/B of the all-layers mixing launches (ops.premix_all) against the per-layer launches on one training step: loss, every parameter's
gradient (bit-identical expected: same kernels' bodies, same order) and the dispatch / timing effect."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import ops  # noqa: E402

dev = torch.device('cuda:0')
mrdis.hip.load()
B, M, H, W = 4, 4, 64, 64
dtype = os.environ.get('CHECK_DTYPE', 'f32')


def run(premix):
    ops._PREMIX = premix
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=['T1', 'T1c', 'T2', 'T2_FLAIR'], input_height=H, input_width=W, batch_size=16, lambda_adv_s=1.0, compute_dtype=dtype)
    cfg = mrdis.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.TrainStep(model, cfg)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=10)
    mask_img = (x[:, 0] == 0).float()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    out = []
    for it in range(3):
        torch.manual_seed(100 + it)
        res = step(xd, mask.to(dev), mask_img.to(dev), mask)
        out.append(float(res['loss']) if isinstance(res, dict) else float(res[0]))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().flatten().float() for p in model.parameters()]).cpu()
    return out, params


l1, p1 = run(True)
l0, p0 = run(False)
print('losses premix   :', l1)
print('losses per-layer:', l0)
print('max |param diff| after 3 steps:', float((p1 - p0).abs().max()), ' identical:', bool(torch.equal(p1, p0)))
