"""Seeded input builders shared by the tests and by the golden-vector generator (oracle/gen_golden.py).

Data only: every function here re-creates an INPUT of a committed fixture from its seed (the fixtures store outputs only).
Nothing in this file reads /root/reference or imports the oracle, so the GPU tests can import it on the GPU box."""
import torch


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def make_inputs(B, M, H, W, seed, drop=False):
    """Synthetic BraTS-shaped batch: N(0,1) inside a centred
    ellipse, -10 outside; mask_img = (inputs[:,0]==0) as util.py builds it."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 7 * M, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    inside = (((yy - H / 2 + 0.5) / (0.40 * H)) ** 2 + ((xx - W / 2 + 0.5) / (0.42 * W)) ** 2) <= 1
    x = torch.where(inside[None, None], x, torch.full_like(x, -10.0))
    mask = torch.ones(B, M)
    if drop:
        for b in range(B):
            d = int(torch.randint(0, M, (1,), generator=g))
            mask[b, d] = 0
            x[b, 7 * d:7 * (d + 1)] = 0
    mask_img = (x[:, 0] == 0).float()
    return x, mask, mask_img


def reinit_discriminator(module, seed=777):
    """Seed-independent deterministic weights for discrim_s so that reference
    and restatement agree without shipping a weight fixture."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in sorted(module.named_parameters()):
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            elif n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.01)


def make_seg_targets(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 4, (B, 1, H, W), generator=g).float()


DATA_CFG = dict(n_subj=6, contrasts=['T1', 'T1c', 'T2', 'T2_FLAIR'], H=40, W=48, D=155, seed=21, missing_every=5)
DATA_SLICES = [0, 1, 2, 3, 77, 100, 151, 150, 149, 120, 30, 64]     # (>= 152 gives the reference a 6-slice item: util.py:483 allows 155-block)


def data_lists():
    subj, idx = [], []
    for s in range(DATA_CFG['n_subj']):
        for k in range(4):
            subj.append(f'BraTS20_Training_{s:03d}'); idx.append(DATA_SLICES[(3 * s + 5 * k) % len(DATA_SLICES)])
    return subj, idx


def dump_measured(name, rec, mode='a'):
    """opt-in record of measured margins: written only when MRDIS_DUMP_MEASURED names a directory (tests do not write into the tree)."""
    import json
    import os
    d = os.environ.get('MRDIS_DUMP_MEASURED')
    if not d:
        return
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), mode) as f:
        f.write(json.dumps(rec) + '\n')
