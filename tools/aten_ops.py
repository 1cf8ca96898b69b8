"""Which ATen ops (torch glue between the HIP kernels) a training step issues: torch.profiler over 2 steps, grouped by op and input
shape, sorted by call count.  HOSTPROF_SHAPE as in host_profile.py (op counts do not depend on the shape).  ATEN_BY_GPU=1: also
record device time and sort by it (run at the bench shape, HOSTPROF_SHAPE=32,4,256,256: which glue ops move the big tensors)."""
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

dev = torch.device('cuda:0')
mrdis.hip.load()
B, M, H, W = (int(v) for v in os.environ.get('HOSTPROF_SHAPE', '4,4,64,64').split(','))
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
N = 2
BY_GPU = os.environ.get('ATEN_BY_GPU', '0') == '1'
with profile(activities=[ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if BY_GPU else []), record_shapes=True) as prof:
    for _ in range(N):
        step(xd, maskd, mimgd, mask)
torch.cuda.synchronize()
VIEWS = {'aten::as_strided', 'aten::slice', 'aten::select', 'aten::view', 'aten::empty', 'aten::t', 'aten::transpose', 'aten::narrow', 'aten::reshape',
         'aten::empty_like', 'aten::empty_strided', 'aten::permute', 'aten::expand', 'aten::unsqueeze', 'aten::squeeze', 'aten::detach', 'aten::alias',
         'aten::contiguous', 'aten::to', 'aten::resize_', 'aten::linear', 'aten::zeros', 'aten::ones', 'aten::pad', 'aten::clone'}
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.key not in VIEWS]
rows.sort(key=lambda e: -(e.self_device_time_total if BY_GPU else e.count))
print(f'{"op":34s} {"calls/step":>10s} {"self cpu us/step":>16s}  input shapes')
for e in rows[:70]:
    print(f'{e.key:34s} {e.count / N:10.1f} {e.self_cpu_time_total / N:16.1f}  ' + (f'gpu {e.self_device_time_total / N:9.1f} us/step  ' if BY_GPU else '') + f'{str(e.input_shapes)[:110]}')
tot = {}
for e in rows:
    t = tot.setdefault(e.key, [0, 0.0]); t[0] += e.count / N; t[1] += e.self_cpu_time_total / N
print('\nby op:')
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f'{k:34s} {c:10.1f} {t:12.1f}')
