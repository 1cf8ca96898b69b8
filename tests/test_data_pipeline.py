"""Batch assembly (reference util.py:444-566, 706-708).

CPU: the oracle restatement against batches produced by the reference's own ZeroDoseDataset + DataLoader
(tests/golden/data_b4.npz, oracle/gen_golden.py data).  GPU: the HBM-resident store + gather kernel against both."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_data as RD
from fixtures import DATA_CFG, data_lists


def _volumes():
    c = DATA_CFG
    return RD.synthetic_volumes(c['n_subj'], c['contrasts'], c['H'], c['W'], c['D'], c['seed'], c['missing_every'])


def _check_batch(b, gold, bi, inputs, targets, mask, mask_img, subj, slices):
    assert list(subj) == list(gold[f'subj_{bi}'])
    np.testing.assert_array_equal(np.asarray(slices), gold[f'slice_{bi}'])
    np.testing.assert_array_equal(mask, gold[f'mask_{bi}'])
    x = inputs.astype(np.float32)
    np.testing.assert_allclose([x.astype(np.float64).sum(), np.abs(x).astype(np.float64).sum()], gold[f'insum_{bi}'], rtol=1e-12)
    np.testing.assert_array_equal(mask_img.sum((1, 2)), gold[f'mimg_{bi}'])
    np.testing.assert_allclose(targets.astype(np.float64).sum((1, 2, 3)), gold[f'tsum_{bi}'], rtol=1e-12)
    if bi < 2:
        np.testing.assert_array_equal(x, gold[f'inputs_{bi}'])
        np.testing.assert_array_equal(targets.astype(np.float32), gold[f'targets_{bi}'])


def test_oracle_batches_match_reference_loader(golden_dir):
    gold = np.load(os.path.join(golden_dir, 'data_b4.npz'))
    data = _volumes()
    subj, idx = data_lists()
    np.random.seed(5); torch.manual_seed(7)
    n = 0
    for bi, b in enumerate(RD.ref_batches('BraTS', data, subj, idx, 4, True, 3, DATA_CFG['contrasts'], True,
                                          (DATA_CFG['H'], DATA_CFG['W']))):
        _check_batch(b, gold, bi, b['inputs'], b['targets'], b['mask'], b['mask_img'], b['subj_id'], b['slice_idx'])
        n += 1
    assert n == int(gold['n_batches'])
    assert any((gold[f'mask_{k}'] == 0).any() for k in range(n))          # missing contrasts and drop-off both occur


@pytest.mark.gpu
def test_device_loader_matches_reference_loader(mrdis, golden_dir):
    """same seeds -> same sample order, same drop-off draws, bit-identical tensors (the kernel only copies)."""
    gold = np.load(os.path.join(golden_dir, 'data_b4.npz'))
    data = _volumes()
    subj, idx = data_lists()
    store = mrdis.VolumeStore.from_arrays(data, 'cuda:0')
    ds = mrdis.SliceDataset('BraTS', store, subj, idx, block_size=3, contrast_list=DATA_CFG['contrasts'], dropoff=True)
    np.random.seed(5); torch.manual_seed(7)
    n = 0
    for bi, b in enumerate(mrdis.BatchLoader(ds, 4, shuffle=True)):
        assert b['inputs'].is_contiguous(memory_format=torch.channels_last)
        _check_batch(b, gold, bi, b['inputs'].cpu().numpy(), b['targets'].cpu().numpy(), b['mask'].cpu().numpy(),
                     b['mask_img'].cpu().numpy(), b['subj_id'], b['slice_idx'].cpu().numpy())
        n += 1
    assert n == int(gold['n_batches'])


@pytest.mark.gpu
def test_device_loader_vs_oracle_unshuffled_other_shape(mrdis):
    """no shuffle, no drop-off, 2 contrasts, odd image size, slices clamped at both ends."""
    data = RD.synthetic_volumes(3, ['T1', 'T2'], 21, 34, 155, seed=3, missing_every=4, with_seg=False)
    subj = [f'BraTS20_Training_{s:03d}' for s in (0, 1, 2, 2, 1, 0, 1)]
    idx = [0, 2, 151, 77, 3, 150, 5]
    store = mrdis.VolumeStore.from_arrays(data, 'cuda:0')
    ds = mrdis.SliceDataset('BraTS', store, subj, idx, block_size=3, contrast_list=['T1', 'T2'], dropoff=False)
    want = list(RD.ref_batches('BraTS', data, subj, idx, 3, False, 3, ['T1', 'T2'], False, (21, 34)))
    got = list(mrdis.BatchLoader(ds, 3, shuffle=False))
    assert len(got) == len(want) == 3
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g['inputs'].cpu().numpy(), w['inputs'])
        np.testing.assert_array_equal(g['mask'].cpu().numpy(), w['mask'])
        np.testing.assert_array_equal(g['mask_img'].cpu().numpy(), w['mask_img'])
        np.testing.assert_array_equal(g['targets'].cpu().numpy(), w['targets'])
        np.testing.assert_array_equal(g['slice_idx'].cpu().numpy(), w['slice_idx'])
    with pytest.raises(IndexError):                                       # the reference's ragged 6-slice item
        ds2 = mrdis.SliceDataset('BraTS', store, subj[:1], [154], block_size=3, contrast_list=['T1', 'T2'])
        next(iter(mrdis.BatchLoader(ds2, 1)))


def test_volume_store_from_h5_walks_the_file_like_the_reference(monkeypatch):
    """VolumeStore.from_h5 (the reference opens its h5 file and indexes it by 'subject/contrast', util.py:455-470, 648-660).  h5py is not in this image, so the
    code path had never executed: here it runs against a stand-in module that implements exactly the h5py surface it uses (File as a context manager,
    visititems over nested groups, Dataset[...]) over nested dicts of arrays -- the walk, the key filter, the (H, W, D) -> (D, H, W) layout, the shape check and the
    missing-h5py error are what is tested; parsing the HDF5 container itself stays h5py's job."""
    import sys
    import types
    import mrdis

    class Dataset:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, idx):
            assert idx is Ellipsis
            return self.a

    class File:
        def __init__(self, path, mode):
            assert mode == 'r'
            self.tree = FILES[path]

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def visititems(self, fn):
            def walk(prefix, node):
                for k, v in node.items():
                    name = f'{prefix}/{k}' if prefix else k
                    if isinstance(v, dict):
                        fn(name, object())                              # a group: not a Dataset
                        walk(name, v)
                    else:
                        fn(name, Dataset(v))
            walk('', self.tree)

    g = np.random.RandomState(3)
    vols = {s: {c: g.randn(6, 5, 9).astype(np.float32) for c in ('T1', 'T2', 'seg')} for s in ('S0', 'S1')}
    FILES = {'brats.h5': vols, 'bad.h5': {'S0': {'T1': np.zeros((6, 5, 9), np.float32), 'T2': np.zeros((6, 5, 8), np.float32)}}}
    stub = types.ModuleType('h5py')
    stub.File, stub.Dataset = File, Dataset
    monkeypatch.setitem(sys.modules, 'h5py', stub)
    st = mrdis.data.VolumeStore.from_h5('brats.h5', 'cpu')
    assert st.shape == (6, 5, 9) and sorted(st.vols) == sorted(f'{s}/{c}' for s in vols for c in vols[s])
    for s in vols:
        for c in vols[s]:
            assert torch.equal(st.vols[f'{s}/{c}'], torch.from_numpy(vols[s][c]).permute(2, 0, 1))      # stored (D, H, W)
    only = mrdis.data.VolumeStore.from_h5('brats.h5', 'cpu', keys={'S1/T2', 'S0/seg'})
    assert sorted(only.vols) == ['S0/seg', 'S1/T2'] and 'S1/T2' in only and 'S1/T1' not in only and only.ptr('S1/T1') == 0
    with pytest.raises(ValueError):
        mrdis.data.VolumeStore.from_h5('bad.h5', 'cpu')                  # volumes of one file share their shape
    monkeypatch.setitem(sys.modules, 'h5py', None)                      # import h5py -> ImportError
    with pytest.raises(RuntimeError, match='h5py is not installed'):
        mrdis.data.VolumeStore.from_h5('brats.h5', 'cpu')
