"""GPU time of the kernels that are NOT the library's (ATen elementwise / cat / reduce, rocBLAS, runtime copies) in one training step at the
bench shape, attributed to the source line that issued them: torch.profiler over one step (after warm-up), every device kernel is charged to its
CPU op; a forward op is named by the innermost frame inside this package, a backward op by its autograd node and the frame of the forward op with
the same sequence number.  Output: a table sorted by GPU time (us per step).  ATEN_SHAPE=B,M,H,W (default 32,4,256,256), ATEN_DTYPE."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrdis  # noqa: E402

dev = torch.device('cuda:0')
mrdis.hip.load()
B, M, H, W = (int(v) for v in os.environ.get('ATEN_SHAPE', '32,4,256,256').split(','))
cfg = dict(mrdis.DEFAULT_CONFIG)
cfg.update(contrast_list=['T1', 'T1c', 'T2', 'T2_FLAIR'][:M], input_height=H, input_width=W, batch_size=B, lambda_adv_s=1.0,
           compute_dtype=os.environ.get('ATEN_DTYPE', 'f32'))
cfg = mrdis.derive_config(cfg, dev)
torch.manual_seed(10); np.random.seed(10)
model = mrdis.build_model(cfg).train()
step = mrdis.TrainStep(model, cfg)
x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=10)
mask_img = (x[:, 0] == 0).float()
xd = x.to(dev).contiguous(memory_format=torch.channels_last)
maskd, mimgd = mask.to(dev), mask_img.to(dev)
torch.manual_seed(100); np.random.seed(100)
for _ in range(3):
    step(xd, maskd, mimgd, mask)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(xd, maskd, mimgd, mask)
    torch.cuda.synchronize()

PKG = os.path.join(ROOT, 'representation-disentanglement_amd')
LIB = ('mrdis', 'wino', 'conv', 'wgrad', 'spade', 'bilinear', 'stat_', 'bn_', 'tapconv', 'pw_', 'recon_', 'lrelu', 'mix_', 'adam', 'sumsq', 'maxpool', 'softmax_md',
       'copy_bytes', 'gn_', 'c4conv', 'instnorm', 'dropoff', 'mailbox', 'cat_', 'gate')


def frame(ev):
    for s in (ev.stack or []):
        if PKG in s and '/hip.py' not in s:
            f = s.replace(PKG + '/', '')
            return f.split(':')[0].strip() if False else f.strip()
    for s in (ev.stack or []):
        if PKG in s:
            return s.replace(PKG + '/', '').strip()
    return '?'


evs = prof.events()
fwd_by_seq = {}
for e in evs:
    if e.device_type.name == 'CPU' and e.sequence_nr is not None and e.sequence_nr >= 0 and e.scope == 0 and e.sequence_nr not in fwd_by_seq:
        if not e.name.startswith('autograd::') and 'Backward' not in e.name:
            fwd_by_seq[e.sequence_nr] = e


def node_of(e):
    p = e
    while p is not None:
        if p.name.startswith('autograd::engine::evaluate_function:'):
            return p
        p = p.cpu_parent
    return None


rows = collections.defaultdict(lambda: [0.0, 0, set()])
lib_us = other_us = 0.0
for e in evs:
    if e.device_type.name != 'CPU' or not e.kernels:
        continue
    if any(c.kernels for c in (e.cpu_children or [])):       # charge a kernel to the innermost op that launched it
        own = [k for k in e.kernels if not any(k in c.kernels for c in e.cpu_children)]
    else:
        own = list(e.kernels)
    for k in own:
        kn = k.name
        us = k.duration
        short = kn.replace('void ', '').replace('at::native::', '')[:60]
        if any(t in kn for t in LIB) and 'at::native' not in kn:
            lib_us += us
            continue
        other_us += us
        nd = node_of(e)
        if nd is not None:
            nm = nd.name.split(': ')[-1]
            f = fwd_by_seq.get(nd.sequence_nr)
            where = f'bwd {nm} <- ' + (frame(f) if f is not None else '?')
        else:
            where = 'fwd ' + frame(e)
        r = rows[(where, e.name)]
        r[0] += us; r[1] += 1; r[2].add(short)

print(f'# one step at B={B} M={M} {H}x{W} {cfg["compute_dtype"]}: library kernels {lib_us / 1e3:.2f} ms, other kernels {other_us / 1e3:.2f} ms')
print(f'# {"us":>8} {"n":>4}  op | where | kernels')
for (where, op), (us, n, ks) in sorted(rows.items(), key=lambda t: -t[1][0])[:int(os.environ.get('ATEN_TOP', '70'))]:
    print(f'{us:10.1f} {n:4d}  {op} | {where} | {"; ".join(sorted(ks))[:90]}')
