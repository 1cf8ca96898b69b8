"""One training step in each compute dtype (f32, bf16m, bf16) against the fp32 reference golden step_b2m4: loss and gradient-norm deviations (the measured bf16 tolerances)."""
import sys; sys.path.insert(0, '/root/repo')
import json, numpy as np, torch, mrdis
from oracle.gen_golden import make_inputs
dev = torch.device('cuda:0')
meta = json.load(open('/root/repo/tests/golden/step_b2m4.json'))
for mode in ('f32', 'bf16m', 'bf16'):
    cfg = dict(mrdis.DEFAULT_CONFIG); cfg.update(contrast_list=[f'm{i}' for i in range(4)], batch_size=16, compute_dtype=mode)
    cfg = mrdis.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.TrainStep(model, cfg)
    inputs, mask, mask_img = make_inputs(2, 4, 160, 192, seed=10)
    torch.manual_seed(11); np.random.seed(11)
    loss, parts, _ = step(inputs.to(dev).contiguous(memory_format=torch.channels_last), mask.to(dev), mask_img.to(dev), mask)
    gn = float(step.last_grad_norm_sq[0].sqrt())
    print(mode, 'loss', float(loss), 'golden', meta['loss'], 'rel', abs(float(loss) - meta['loss']) / meta['loss'], 'gnorm', gn, meta['grad_norm'],
          {k: round(float(v), 5) for k, v in parts.items() if float(v) != 0}, flush=True)
