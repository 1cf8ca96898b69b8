"""Where the small ATen ops of a training step's FORWARD come from (their autograd counterparts run on the engine's thread and roughly mirror
them): a TorchDispatchMode records every non-view aten op with the innermost frame inside this package, over one step at a small shape."""
import collections
import os
import sys
import traceback

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

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
VIEWS = ('as_strided', 'slice', 'select', 'view', 'empty', 't.', 'transpose', 'narrow', 'reshape', 'permute', 'expand', 'unsqueeze', 'squeeze', 'detach',
         'alias', '_unsafe_view', 'unfold', 'resize_', 'lift_fresh', 'is_', '_local_scalar', 'sym_', 'stride', 'size', 'numel', 'dim', 'record_stream',
         'split', 'chunk', 'unbind', 'contiguous', '_to_copy', 'to.', 'flatten', 'empty_like', 'empty_strided', 'new_empty', 'set_', 'item', 'is_pinned', 'pin_memory')
agg = collections.Counter(); ops_at = collections.defaultdict(collections.Counter); big = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if not any(name.startswith(v) for v in VIEWS) and any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())):
            where = '(no package frame)'
            for fr in reversed(traceback.extract_stack(limit=40)):
                if 'representation-disentanglement_amd' in fr.filename and 'aten_sources' not in fr.filename and fr.name not in ('nhwc', 'cast_view'):
                    where = f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}'
                    break
            agg[where] += 1; ops_at[where][name.split('.')[0]] += 1
            out = func(*args, **(kwargs or {}))
            o = out[0] if isinstance(out, (tuple, list)) and out else out
            if isinstance(o, torch.Tensor) and o.numel() * o.element_size() >= (8 << 20):
                big[(where, name.split('.')[0], tuple(o.shape), str([tuple(a.stride()) for a in args if isinstance(a, torch.Tensor)][:2]))] += 1
            return out
        return func(*args, **(kwargs or {}))


with Rec():
    step(xd, maskd, mimgd, mask)
torch.cuda.synchronize()
print(f'{sum(agg.values())} non-view ATen ops on device tensors in one step (main thread: forward + optimizer)')
for w, c in agg.most_common(50):
    print(f'{c:6d}  {w[:70]:70s} ' + ', '.join(f'{k} {v}' for k, v in ops_at[w].most_common(5)))
print('ATen ops whose result is >= 8 MiB (source, op, result shape, argument strides): count')
for k, c in sorted(big.items(), key=lambda kv: -kv[1] * 1)[:40]:
    print(f'{c:4d}  {k[0][:50]:50s} {k[1]:12s} {k[2]}  {k[3]}')
