"""Batch assembly: HBM-resident store + gather kernel vs the CPU restatement of the reference's loader
(oracle/ref_data.py: numpy slicing / concatenate / collate, then the H2D copy the training loop does).
BraTS geometry: 160x192x155 volumes, 4 contrasts, block 3, batch 32."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from oracle import ref_data as RD  # noqa: E402


def main():
    n_subj, B = 12, 32
    contrasts = ['T1', 'T1c', 'T2', 'T2_FLAIR']
    H, W, D = 160, 192, 155
    data = RD.synthetic_volumes(n_subj, contrasts, H, W, D, seed=1, with_seg=True)
    rng = np.random.RandomState(0)
    subj = [f'BraTS20_Training_{rng.randint(n_subj):03d}' for _ in range(8 * B)]
    idx = [int(rng.randint(3, 152)) for _ in range(8 * B)]
    dev = torch.device('cuda:0')
    t0 = time.perf_counter()
    store = mrdis.VolumeStore.from_arrays(data, dev)
    torch.cuda.synchronize()
    gb = sum(v.numel() for v in store.vols.values()) * 4 / 1e9
    print(f'store: {len(store.vols)} volumes, {gb:.2f} GB resident, uploaded in {time.perf_counter() - t0:.1f} s')
    ds = mrdis.SliceDataset('BraTS', store, subj, idx, block_size=3, contrast_list=contrasts, dropoff=True)
    loader = mrdis.BatchLoader(ds, B, shuffle=True)
    for _ in loader:                                    # warm-up
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    for b in loader:
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    bytes_b = B * (len(contrasts) * 7 + 1) * H * W * 4 * 2          # read + write of inputs and targets
    print(f'device loader: {dt * 1e3:.2f} ms/batch end to end ({B / dt:.0f} slices/s); gather traffic {bytes_b / 1e6:.0f} MB/batch')
    # kernel-only time
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    metas = [ds.meta(i) for i in range(B)]
    ptrs = torch.tensor([m[2] for m in metas], dtype=torch.int64, device=dev)
    sl = torch.tensor([m[1] for m in metas], dtype=torch.int32, device=dev)
    dr = torch.full((B,), -1, dtype=torch.int32, device=dev)
    mrdis.hip.slice_gather(ptrs, sl, dr, H, W, D, 3); torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        mrdis.hip.slice_gather(ptrs, sl, dr, H, W, D, 3)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mb = B * len(contrasts) * 7 * H * W * 4 * 2 / 1e6
    print(f'slice_gather kernel: {us:.1f} us for {mb:.0f} MB read+written = {mb / us:.2f} TB/s')
    # CPU loader of the reference (restated), incl. the H2D copy of main_missing.py:157-161
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.perf_counter(); n = 0
    for b in RD.ref_batches('BraTS', data, subj, idx, B, True, 3, contrasts, True, (H, W)):
        x = torch.from_numpy(b['inputs']).to(dev); m = torch.from_numpy(b['mask']).to(dev); mi = torch.from_numpy(b['mask_img']).to(dev)
        n += 1
        if n == 4:
            break
    torch.cuda.synchronize()
    dtc = (time.perf_counter() - t0) / n
    print(f'CPU loader (restated reference, in-memory volumes, + H2D): {dtc * 1e3:.1f} ms/batch ({B / dtc:.0f} slices/s)')


if __name__ == '__main__':
    main()
