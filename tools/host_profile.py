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
B, M, H, W = (int(v) for v in os.environ.get('HOSTPROF_SHAPE', '32,4,256,256').split(','))      # a tiny shape shows the pure host cost
cfg = dict(mrdis.DEFAULT_CONFIG)
cfg.update(contrast_list=['T1', 'T1c', 'T2', 'T2_FLAIR'], input_height=H, input_width=W, batch_size=max(B, 16), lambda_adv_s=1.0,
           compute_dtype=os.environ.get('HOSTPROF_DTYPE', 'f32'))
cfg = mrdis.derive_config(cfg, dev)
torch.manual_seed(10); np.random.seed(10)
model = mrdis.build_model(cfg).train()
step = mrdis.TrainStep(model, cfg)
x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=10)
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
import time
t0 = time.perf_counter()
for _ in range(5):
    step(xd, maskd, mimgd, mask)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'un-profiled: host enqueue {(t1 - t0) / 5 * 1e3:.1f} ms/step, with final sync {(t2 - t0) / 5 * 1e3:.1f} ms/step')
st.sort_stats('tottime').print_stats(45)
