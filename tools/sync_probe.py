"""Does a training step synchronise the host with the GPU anywhere?  Two steps under torch.cuda.set_sync_debug_mode('warn'): every synchronising torch call is listed with
its source line (r04: none -- the loss scalars travel once per logging interval, the small host -> device transfers through the pinned mailbox)."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import mrdis
dev = torch.device('cuda:0'); mrdis.hip.load()
B, M, H, W = 8, 4, 128, 128
cfg = dict(mrdis.DEFAULT_CONFIG); cfg.update(contrast_list=['T1','T1c','T2','T2_FLAIR'], input_height=H, input_width=W, batch_size=16, lambda_adv_s=1.0)
cfg = mrdis.derive_config(cfg, dev)
torch.manual_seed(10); np.random.seed(10)
model = mrdis.build_model(cfg).train(); step = mrdis.TrainStep(model, cfg)
x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=10)
mask_img = (x[:, 0] == 0).float()
xd = x.to(dev).contiguous(memory_format=torch.channels_last); maskd, mimgd = mask.to(dev), mask_img.to(dev)
for _ in range(3): step(xd, maskd, mimgd, mask)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as wlist:
    warnings.simplefilter('always')
    for _ in range(2): step(xd, maskd, mimgd, mask)
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
print('synchronising calls in two steps:', len(wlist))
for w in wlist[:20]:
    print(' ', str(w.message)[:150], '|', w.filename.split('/')[-1], w.lineno)
