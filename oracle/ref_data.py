"""Test oracle for the batch assembly (NOT product code; imported only by tests/ and tools/data_bench.py's baseline).

CPU restatement of `ZeroDoseDataset.__getitem__` (src/util.py:471-566) and of the DataLoader around it
(util.py:706-708, num_workers = 0): numpy slicing / concatenation and torch's default collate.  Pinned by
`tests/golden/data_*.npz`, produced by `oracle/gen_golden.py data` from the reference's own class fed with a dict of
numpy volumes (the class only needs `key in data` and `data[key][:, :, a:b]`).
"""
import numpy as np
import torch


def ref_getitem(dataset_name, data, subj_id, slice_idx, block_size, contrast_list, dropoff, image_size):
    s = int(slice_idx)
    if s < block_size:                                                      # :476-484
        s = block_size
    hi = (89 if dataset_name == 'Tau' else 155) - block_size
    if s > hi:
        s = hi
    imgs, mask = [], []
    for c in contrast_list:                                                 # :519-525
        key = subj_id + '/' + c
        if key in data:
            imgs.append(data[key][:, :, s - block_size:s + block_size + 1]); mask.append(1)
        else:
            imgs.append(np.zeros((image_size[0], image_size[1], 2 * block_size + 1))); mask.append(0)
    mask = np.array(mask)
    inputs = np.concatenate(imgs, 2)
    if dataset_name == 'BraTS' and subj_id + '/seg' in data:                # :531-535
        targets = np.array(data[subj_id + '/seg'][:, :, s:s + 1])
        targets[targets == 4] = 3.
    elif dataset_name == 'ZeroDose' and subj_id + '/PET' in data:
        targets = data[subj_id + '/PET'][:, :, s:s + 1]
    else:
        targets = np.zeros((image_size[0], image_size[1], 1))
    if dropoff and mask.sum() > 1:                                          # :538-542
        if np.random.rand() > 0.8:
            drop_idx = np.random.choice(np.where(mask == 1)[0], 1)[0]
            c7 = 2 * block_size + 1
            inputs[:, :, drop_idx * c7:(drop_idx + 1) * c7] = 0
            mask[drop_idx] = 0
    inputs = np.transpose(inputs, (2, 0, 1))
    targets = np.transpose(targets, (2, 0, 1))
    mask_img = (inputs[0] == 0).astype(float)                               # :563
    return {'inputs': inputs, 'targets': targets, 'subj_id': subj_id, 'slice_idx': s, 'mask': mask, 'mask_img': mask_img}


def ref_batches(dataset_name, data, subj_list, idx_list, batch_size, shuffle, block_size, contrast_list, dropoff, image_size):
    """generator of collated batches in DataLoader order (RandomSampler semantics for shuffle)."""
    n = len(subj_list)
    if shuffle:
        torch.empty((), dtype=torch.int64).random_()          # DataLoader's _base_seed draw comes first (dataloader.py, _BaseDataLoaderIter)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        g = torch.Generator(); g.manual_seed(seed)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    for i0 in range(0, n, batch_size):
        items = [ref_getitem(dataset_name, data, str(subj_list[i]), idx_list[i], block_size, contrast_list, dropoff, image_size)
                 for i in order[i0:i0 + batch_size]]
        yield {'inputs': np.stack([it['inputs'] for it in items]).astype(np.float32),
               'targets': np.stack([it['targets'] for it in items]).astype(np.float32),
               'subj_id': [it['subj_id'] for it in items],
               'slice_idx': np.array([it['slice_idx'] for it in items]),
               'mask': np.stack([it['mask'] for it in items]).astype(np.float32),
               'mask_img': np.stack([it['mask_img'] for it in items]).astype(np.float32)}


def synthetic_volumes(n_subj, contrasts, H, W, D, seed, missing_every=0, with_seg=True):
    """small BraTS-shaped store: z-scored values inside an ellipse, exact zeros outside (mask_img needs real zeros)."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    inside = (((yy - H / 2 + 0.5) / (0.4 * H)) ** 2 + ((xx - W / 2 + 0.5) / (0.42 * W)) ** 2) <= 1
    data = {}
    for s in range(n_subj):
        sid = f'BraTS20_Training_{s:03d}'
        for ci, c in enumerate(contrasts):
            if missing_every and (s + ci) % missing_every == 0:
                continue
            v = rng.randn(H, W, D).astype(np.float32)
            v[~inside] = 0.0
            data[sid + '/' + c] = v
        if with_seg:
            data[sid + '/seg'] = (rng.randint(0, 5, size=(H, W, D)) * inside[:, :, None]).astype(np.float32)
    return data
