"""Where the HOST time of one training step goes (cProfile over 3 steps of the bench configuration)."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

dev = torch.device('cuda:0')
mrdis.hip.load()
B, M, H, W = 32, 4, 256, 256
cfg = dict(mrdis.DEFAULT_CONFIG)
cfg.update(contrast_list=['T1', 'T1c', 'T2', 'T2_FLAIR'], input_height=H, input_width=W, batch_size=B, lambda_adv_s=1.0)
cfg = mrdis.derive_config(cfg, dev)
torch.manual_seed(10); np.random.seed(10)
model = mrdis.build_model(cfg).train()
step = mrdis.TrainStep(model, cfg)
x, mask, mask_img = mrdis.synthetic_batch(B, M, 240, 240, seed=10)
x = mrdis.fit_to_model(x, (H, W), fill=-10.0)
mask_img = (x[:, 0] == 0).float()
xd = x.to(dev).contiguous(memory_format=torch.channels_last)
maskd, mimgd = mask.to(dev), mask_img.to(dev)
for _ in range(2):
    step(xd, maskd, mimgd, mask)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step(xd, maskd, mimgd, mask)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
