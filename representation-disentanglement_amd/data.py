"""Batch assembly of the reference's slice dataset (`src/util.py:444-566, 635-721`) with the volumes resident in HBM.

The reference keeps the volumes in an h5 file laid out (H, W, D) per `subject/contrast`, and every
`__getitem__` reads 2*block+1 slices per contrast, concatenates, transposes and returns numpy arrays that the
DataLoader collates and the training loop copies to the GPU (110 MB per B=32 batch).  One MI355X holds a whole
BraTS fold (~28 GB fp32) in 10 % of its HBM, so here the volumes are uploaded once (as [D][H][W] planes) and a
batch is ONE gather kernel driven by three small index arrays (`mrdis_slice_gather`).

Host-side behaviour follows the reference statement by statement:
  * slice clamping to [block, 155-block] (89-block for 'Tau')                     util.py:476-484
  * a contrast missing for a subject gives zeros and mask 0                         util.py:519-525
  * drop-off: `np.random.rand() > 0.8` then `np.random.choice(present, 1)` per item, only when dropoff is
    set and more than one contrast is present -- same global-RNG call order        util.py:538-542
  * mask_img = (inputs[0] == 0)                                                     util.py:563
  * BraTS targets: the `seg` slice with label 4 -> 3                                util.py:531-535
  * batch order: `DataLoader(shuffle)` = `RandomSampler` seeded from torch's default generator
    (num_workers = 0, main_missing.py:63), reproduced draw for draw.
`skull_strip` and the `aug` flip (a `pdb.set_trace()` in the reference, util.py:555-558) are not on the path.
"""
import numpy as np
import torch

from . import hip


class VolumeStore:
    """`data[subject + '/' + contrast]` of the reference's h5 file, kept on the device as (D, H, W)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.vols = {}
        self.shape = None           # (H, W, D) of every volume

    def add(self, key, array_hwd):
        """key = 'subject/contrast' (h5 path); array in the reference layout (H, W, D)."""
        a = torch.as_tensor(np.ascontiguousarray(array_hwd), dtype=torch.float32)
        if a.dim() != 3:
            raise ValueError('volume must be (H, W, D)')
        if self.shape is None:
            self.shape = tuple(a.shape)
        elif tuple(a.shape) != self.shape:
            raise ValueError(f'{key}: shape {tuple(a.shape)} != {self.shape}')
        self.vols[key] = a.permute(2, 0, 1).contiguous().to(self.device)

    @classmethod
    def from_arrays(cls, arrays, device):
        st = cls(device)
        for k, v in arrays.items():
            st.add(k, v)
        return st

    @classmethod
    def from_h5(cls, path, device, keys=None):
        """the reference's file (util.py:643-694); needs h5py, which this image does not ship."""
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError('h5py is not installed: build the store with VolumeStore.from_arrays') from e
        st = cls(device)
        with h5py.File(path, 'r') as f:
            def visit(name, obj):
                if isinstance(obj, h5py.Dataset) and (keys is None or name in keys):
                    st.add(name, obj[...])
            f.visititems(visit)
        return st

    def __contains__(self, key):
        return key in self.vols

    def ptr(self, key):
        v = self.vols.get(key)
        return 0 if v is None else v.data_ptr()


def load_idx_list(path):
    """util.py:717-719: 'subject slice' lines, space separated, no header."""
    import pandas as pd
    lines = pd.read_csv(path, sep=' ', header=None)
    return np.array(lines.iloc[:, 0]), np.array(lines.iloc[:, 1])


class SliceDataset:
    """ZeroDoseDataset (util.py:444-566) over a VolumeStore: `meta(idx)` is the host part of `__getitem__`."""

    def __init__(self, dataset_name, store, subj_list, idx_list, block_size=3, contrast_list=('T1',), dropoff=False):
        self.dataset_name, self.store = dataset_name, store
        self.subj_list, self.idx_list = list(subj_list), list(idx_list)
        self.block_size, self.contrast_list, self.dropoff = block_size, list(contrast_list), dropoff

    def __len__(self):
        return len(self.subj_list)

    def meta(self, idx, rng=None):
        """rng: where the drop-off draws come from -- None = the global np.random (the reference's own stream, util.py:538-542: single-process runs reproduce
        it draw for draw), or a np.random.RandomState of the loader's own (data-parallel runs, see BatchLoader)."""
        rng = np.random if rng is None else rng
        subj_id = str(self.subj_list[idx])
        s = int(self.idx_list[idx])
        b = self.block_size
        if s < b:                                                    # :476-484
            s = b
        hi = (89 if self.dataset_name == 'Tau' else 155) - b
        if s > hi:
            s = hi
        if s + b > self.store.shape[2] - 1:
            # the reference's upper clamp (155 - block) still lets slice 152 of a 155-slice volume through and then returns a
            # 6-slice item that its DataLoader cannot collate; its index lists never contain such slices
            raise IndexError(f'slice {s} +/- {b} leaves the volume (D = {self.store.shape[2]})')
        ptrs = [self.store.ptr(subj_id + '/' + c) for c in self.contrast_list]
        mask = np.array([1 if p else 0 for p in ptrs])
        drop = -1
        if self.dropoff and mask.sum() > 1:                           # :538-542, same RNG call order
            if rng.rand() > 0.8:
                drop = int(rng.choice(np.where(mask == 1)[0], 1)[0])
        target_key = {'ZeroDose': '/PET', 'BraTS': '/seg', 'Tau': '/pet_nifti/fulldose'}.get(self.dataset_name)
        tptr = self.store.ptr(subj_id + target_key) if target_key else 0
        return subj_id, s, ptrs, drop, tptr


class BatchLoader:
    """DataLoader(dataset, batch_size, shuffle, num_workers=0) of util.py:706-708; yields the reference's sample dict
    with device tensors: inputs (B, 7M, H, W) channels_last, targets (B, 1, H, W), mask (B, M), mask_img (B, H, W),
    subj_id (list), slice_idx (B,).

    Data parallel (`world` > 1; the reference is single-device, main_missing.py:28): every rank walks the SAME batch sequence
    -- the permutation and the per-item drop-off draws of ALL batches are made on every rank -- and rank r assembles batches
    r, r + world, r + 2 world, ...  The drop-off draws then come from a np.random.RandomState OF THE LOADER (seeded once, from the
    global np.random at the first epoch: identical on every rank because the entry point seeds every rank alike), not from the global
    np.random: the model draws from the global stream inside every training step (the sim_s / adversarial pair picks, model.py:3487,
    :3567), and a rank consumes the metas of `world` batches between two steps, so on the shared stream the masks of batch k would
    depend on the rank and on `world`.  With the loader's own stream every rank sees the same masks for batch k and the global stream
    carries only the model's draws (identical on every rank).  world == 1 keeps the reference's single stream, draw for draw
    (tests/golden/data_b4.npz).  `limit` (batches(limit)): stop after that many GLOBAL batches on every rank -- the reference's
    evaluation loop breaks after batch 501 (main_missing.py:562-563); no meta of a later batch is drawn on any rank.
    `equal_steps` (the train loader): the ragged tail is dropped -- incomplete batches, then whole batches beyond the last full
    round of `world` -- so every rank runs the same number of optimizer steps on full batches (the gradient exchange is a
    collective; BatchNorm and the roll-by-one negatives of sim_s / sim_z want B >= 2 on every rank).  Without it (val / test
    loaders: no collective inside a batch) every batch is served, ranks may differ by one batch.  `generator`: a
    torch.Generator the permutation seeds are drawn from instead of the default one (the entry point hands every rank an
    identically seeded one, so that the default generator can be seeded per rank for the noise of `sample`)."""

    def __init__(self, dataset, batch_size, shuffle=False, rank=0, world=1, equal_steps=False, generator=None):
        if not 0 <= rank < world:
            raise ValueError(f'rank {rank} outside world {world}')
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rank, self.world, self.equal_steps, self.generator = rank, world, equal_steps, generator
        self._drop_rng = None                          # world > 1: the loader's own drop-off stream (see the class docstring)

    def global_batches(self):
        """number of batches all ranks walk together per epoch."""
        n, bs = len(self.dataset), self.batch_size
        if self.world > 1 and self.equal_steps:
            return n // bs // self.world * self.world
        return (n + bs - 1) // bs

    def __len__(self):
        g = self.global_batches()
        return (g - self.rank + self.world - 1) // self.world if g > self.rank else 0

    def _order(self):
        n = len(self.dataset)
        if not self.shuffle:
            return list(range(n))
        # DataLoader.__iter__ draws its `_base_seed` from the default generator first (used only by worker processes),
        # then RandomSampler.__iter__ (generator=None) draws the seed of the permutation on the first next()
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        seed = int(torch.empty((), dtype=torch.int64).random_(generator=self.generator).item())
        g = torch.Generator()
        g.manual_seed(seed)
        return torch.randperm(n, generator=g).tolist()

    def batch_plan(self, limit=None):
        """host half of an epoch: yields (global batch index, dataset indices, item metas) of THIS rank's batches (of the first `limit` global batches)."""
        order = self._order()
        ds = self.dataset
        rng = None
        if self.world > 1:
            if self._drop_rng is None:
                self._drop_rng = np.random.RandomState(int(np.random.randint(0, 2 ** 31 - 1)))      # one draw from the (identically seeded) global stream, on every rank
            rng = self._drop_rng
        g = self.global_batches()
        for k in range(g if limit is None else min(g, limit)):
            idxs = order[k * self.batch_size:(k + 1) * self.batch_size]
            metas = [ds.meta(i, rng) for i in idxs]                   # every rank draws the metas of ALL batches: the streams stay in step
            if k % self.world == self.rank:
                yield k, idxs, metas

    def __iter__(self):
        return self.batches()

    def batches(self, limit=None):
        ds, st = self.dataset, self.dataset.store
        H, W, D = st.shape
        M, blk = len(ds.contrast_list), ds.block_size
        dev = st.device
        for k, _, metas in self.batch_plan(limit):
            B = len(metas)
            host = torch.empty((B, M + 3), dtype=torch.int64).pin_memory() if dev.type == 'cuda' else torch.empty((B, M + 3), dtype=torch.int64)
            mask_host = np.zeros((B, M), dtype=np.float32)                 # host twin of `mask`: the losses branch on it without a sync
            for r, (_, s, ptrs, drop, tptr) in enumerate(metas):
                host[r, :M] = torch.tensor(ptrs, dtype=torch.int64)
                host[r, M], host[r, M + 1], host[r, M + 2] = s, drop, tptr
                mask_host[r] = [1.0 if (p and m != drop) else 0.0 for m, p in enumerate(ptrs)]
            d = host.to(dev, non_blocking=True)                       # one small H2D copy per batch
            vol_ptrs = d[:, :M].contiguous()
            slice_idx = d[:, M].to(torch.int32)
            drop = d[:, M + 1].to(torch.int32)
            inputs, mask, mask_img = hip.slice_gather(vol_ptrs, slice_idx, drop, H, W, D, blk)
            none = torch.full((B,), -1, dtype=torch.int32, device=dev)
            targets, _, _ = hip.slice_gather(d[:, M + 2:M + 3].contiguous(), slice_idx, none, H, W, D, 0)
            if ds.dataset_name == 'BraTS':
                targets = torch.where(targets == 4, torch.full_like(targets, 3.0), targets)      # util.py:533
            yield {'inputs': inputs, 'targets': targets, 'subj_id': [m[0] for m in metas],
                   'slice_idx': slice_idx.to(torch.int64), 'mask': mask, 'mask_img': mask_img, 'mask_host': mask_host,
                   'batch_index': k}
