"""The entry point as a data-parallel job (north_star; BASELINE configs[2] / [3]): the host logic of mrdis/train.py and
mrdis/data.py for WORLD_SIZE > 1, on CPU with gloo.  The reference is single-device (main_missing.py:28, :62-68, :141-164,
:307-335); what is pinned here is that N ranks behave like ONE run of it: the train loader's shards are disjoint, equally
long and their union is the single-process batch sequence; epoch means / the validation monitor are the means over all
ranks, so ReduceLROnPlateau takes the same decision everywhere; files are written once, by rank 0."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _dataset(mrdis, n_items=37, dropoff=True):
    H, W, D = 8, 8, 12
    g = np.random.RandomState(5)
    arrays = {}
    for s in range(3):
        for c in ('T1', 'T2'):
            if (s, c) != (2, 'T2'):                                  # one subject lacks a contrast (util.py:519-525)
                arrays[f'S{s}/{c}'] = g.randn(H, W, D).astype(np.float32)
    store = mrdis.data.VolumeStore.from_arrays(arrays, 'cpu')
    subj = [f'S{i % 3}' for i in range(n_items)]
    idx = [3 + i % 5 for i in range(n_items)]
    return mrdis.data.SliceDataset('BraTS', store, subj, idx, block_size=3, contrast_list=['T1', 'T2'], dropoff=dropoff)


def _epoch(loader, seed):
    torch.manual_seed(seed); np.random.seed(seed)
    return [(k, idxs, [m[3] for m in metas]) for k, idxs, metas in loader.batch_plan()]


@pytest.mark.parametrize('world', [2, 3, 8])
def test_train_loader_shards_are_disjoint_equal_and_cover_the_single_process_order(world):
    import mrdis
    ds = _dataset(mrdis)
    bs = 2
    single = _epoch(mrdis.data.BatchLoader(ds, bs, shuffle=True), 7)
    assert len(single) == 19 and len(single[-1][1]) == 1              # 37 items: 18 full batches + a ragged one
    shards = [_epoch(mrdis.data.BatchLoader(ds, bs, shuffle=True, rank=r, world=world, equal_steps=True), 7) for r in range(world)]
    rounds = 18 // world
    assert all(len(s) == rounds for s in shards)                      # the same number of optimizer steps everywhere
    for r, s in enumerate(shards):
        assert [k for k, _, _ in s] == list(range(r, rounds * world, world))
        assert len(mrdis.data.BatchLoader(ds, bs, shuffle=True, rank=r, world=world, equal_steps=True)) == rounds
    seen = [i for s in shards for _, idxs, _ in s for i in idxs]
    assert len(seen) == len(set(seen))                                # disjoint
    merged = sorted((b for s in shards for b in s), key=lambda b: b[0])
    assert [(k, idxs) for k, idxs, _ in merged] == [(k, idxs) for k, idxs, _ in single[:rounds * world]]      # union = the single-process batch sequence
    # the drop-off draws of a data-parallel run come from the loader's own stream (seeded by ONE draw from the identically seeded global stream):
    # what every rank must see for batch k, computed independently here
    torch.manual_seed(7); np.random.seed(7)
    probe = mrdis.data.BatchLoader(ds, bs, shuffle=True, rank=0, world=world, equal_steps=True)
    order = probe._order()
    rs = np.random.RandomState(int(np.random.randint(0, 2 ** 31 - 1)))
    want = [[ds.meta(i, rs)[3] for i in order[k * bs:(k + 1) * bs]] for k in range(rounds * world)]
    assert [drops for _, _, drops in merged] == want
    assert any(d >= 0 for drops in want for d in drops)               # the fixture does drop modalities
    # ... and they do not depend on what the MODEL draws from the global stream between two batches (sim_s / adversarial pair picks, model.py:3487, :3567):
    # a rank takes the metas of `world` batches between two steps, so on a shared stream its masks would depend on rank and world
    for r in range(world):
        torch.manual_seed(7); np.random.seed(7)
        noisy = []
        for k, idxs, metas in mrdis.data.BatchLoader(ds, bs, shuffle=True, rank=r, world=world, equal_steps=True).batch_plan():
            noisy.append((k, idxs, [m[3] for m in metas]))
            np.random.choice(8, 2, replace=False)                     # the model's draw inside the training step
        assert noisy == shards[r]
    # the host generators end the epoch in the same state on every rank (next epoch's permutation, the sim_s pair draws)
    states = []
    for r in range(world):
        _epoch(mrdis.data.BatchLoader(ds, bs, shuffle=True, rank=r, world=world, equal_steps=True), 7)
        states.append((float(np.random.rand()), int(torch.empty((), dtype=torch.int64).random_())))
    assert all(s == states[0] for s in states)


@pytest.mark.parametrize('world', [2, 8])
def test_evaluation_cap_is_on_the_global_batch_index(world):
    """main_missing.py:562-563 stops the evaluation loop after batch 501; under data parallelism every rank must stop at the same GLOBAL batch (a per-rank test
    after the step let world = 8 run up to batch 508) and leave the host generators in the same state (no meta of a later batch is drawn anywhere)."""
    import mrdis
    ds = _dataset(mrdis, n_items=61, dropoff=True)
    cap = 13
    seen, states = [], []
    for r in range(world):
        torch.manual_seed(5); np.random.seed(5)
        got = [k for k, _, _ in mrdis.data.BatchLoader(ds, 2, rank=r, world=world).batch_plan(limit=cap)]
        assert got == list(range(r, cap, world))
        seen += got
        states.append(float(np.random.rand()))
    assert sorted(seen) == list(range(cap)) and all(s == states[0] for s in states)
    torch.manual_seed(5); np.random.seed(5)
    assert [k for k, _, _ in mrdis.data.BatchLoader(ds, 2).batch_plan(limit=cap)] == list(range(cap))      # single process: batches 0 .. cap - 1, as the reference


def test_eval_loader_serves_every_batch_once():
    import mrdis
    ds = _dataset(mrdis, n_items=23, dropoff=False)
    single = _epoch(mrdis.data.BatchLoader(ds, 4), 1)
    assert len(single) == 6 and len(single[-1][1]) == 3
    shards = [_epoch(mrdis.data.BatchLoader(ds, 4, rank=r, world=4), 1) for r in range(4)]
    assert [len(s) for s in shards] == [2, 2, 1, 1]
    assert [len(mrdis.data.BatchLoader(ds, 4, rank=r, world=4)) for r in range(4)] == [2, 2, 1, 1]
    assert sorted((b for s in shards for b in s), key=lambda b: b[0]) == single
    with pytest.raises(ValueError):
        mrdis.data.BatchLoader(ds, 4, rank=4, world=4)


def test_single_process_loader_is_unchanged():
    """world = 1 must stay the reference's DataLoader draw for draw (tests/golden/data_b4.npz pins the tensors on the GPU)."""
    import mrdis
    ds = _dataset(mrdis, n_items=11)
    torch.manual_seed(3); np.random.seed(3)
    a = [(idxs, [m[3] for m in metas]) for _, idxs, metas in mrdis.data.BatchLoader(ds, 4, shuffle=True).batch_plan()]
    torch.manual_seed(3); np.random.seed(3)
    torch.empty((), dtype=torch.int64).random_()
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator(); g.manual_seed(seed)
    order = torch.randperm(11, generator=g).tolist()
    assert [i for idxs, _ in a for i in idxs] == order
    assert len(a) == 3 and len(a[-1][0]) == 3


# --------------------------------------------------------------------------- Run.train over two gloo ranks
class _PlanLoader:
    """the host half of BatchLoader (no gather kernel): yields the planned batches as sample dicts."""

    def __init__(self, loader):
        self.loader = loader

    def __iter__(self):
        for k, idxs, metas in self.loader.batch_plan():
            yield {'inputs': None, 'mask': None, 'mask_img': None, 'mask_host': None, 'targets': None, 'batch_index': k, 'idxs': idxs}


def _entry_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      LOCAL_WORLD_SIZE=str(world))
    import yaml
    import mrdis
    T = mrdis.train
    try:
        over = T.init_distributed({}, backend='gloo')
        assert over == {'gpu': str(rank)} and T.dist_info() == (rank, world)
        cfgfile = os.path.join(root, 'config.yaml')
        if rank == 0:
            with open(cfgfile, 'w') as f:
                yaml.dump(dict(contrast_list=['T1', 'T2'], input_height=32, input_width=32, batch_size=2, epochs=8,
                               ckpt_root=os.path.join(root, 'ckpt'), dropoff=True), f)
        dist.barrier()
        if rank != 0:                                                  # every file of the run comes from rank 0
            def boom(*a, **k):
                raise AssertionError('a rank other than 0 wrote a file')
            T.save_checkpoint = T.save_result_stat = T.save_config_file = T.save_config_yaml = boom
        config = T.setup_config(cfgfile, device=torch.device('cpu'))
        paths = [None] * world
        dist.all_gather_object(paths, config['ckpt_path'])
        assert len(set(paths)) == 1                                    # one directory per job (rank 0's clock names it)
        ds = _dataset(mrdis)
        loaders = {'train': _PlanLoader(mrdis.data.BatchLoader(ds, 2, shuffle=True, rank=rank, world=world, equal_steps=True,
                                                               generator=torch.Generator().manual_seed(T.SEED))),
                   'val': _PlanLoader(mrdis.data.BatchLoader(_dataset(mrdis, 6, dropoff=False), 2, rank=rank, world=world))}
        torch.manual_seed(0)
        run = T.Run(config, loaders=loaders, log=lambda *a: None, model=nn.Linear(3, 2))
        assert run.step.reducer is not None and run.step.reducer.world == world
        nk = len(mrdis.LOSS_KEYS)
        trained = []

        def fake_step(inputs, mask, mask_img, mask_host=None, targets=None, it=None):
            trained.append(list(loaders_seen[-1]))
            v = float(sum(loaders_seen[-1]))                           # a loss that depends on WHICH items this rank got
            return torch.tensor(v), {k: torch.tensor(v + j) for j, k in enumerate(mrdis.LOSS_KEYS)}, None

        calls = {'val': 0}

        def fake_eval(inputs, mask, mask_img, mask_host=None, targets=None):
            # val has 3 batches: rank 0 serves two, rank 1 one.  Rank 0 alone sees an improving monitor, rank 1 a worsening one; the mean
            # over the three batches is CONSTANT, so the run plateaus -- only if the monitor is reduced before scheduler.step
            calls['val'] += 1
            epoch = (calls['val'] - 1) // (2 if rank == 0 else 1)
            lrs[epoch] = run.optimizer.lr                              # the rate in effect during epoch `epoch`
            v = 1.0 - 0.01 * epoch if rank == 0 else 1.0 + 0.02 * epoch
            parts = {k: torch.tensor(v) for k in mrdis.LOSS_KEYS}
            met = {'rmse': torch.full((2,), float(rank)), 'psnr': torch.full((2,), 20.0 + rank), 'ssim': torch.full((2,), 0.5)}
            return torch.tensor(v), parts, met, None
        # remember the indices of the batch the loop is about to hand to the step
        loaders_seen = []
        orig_iter = _PlanLoader.__iter__

        def tracking_iter(self):
            for sample in orig_iter(self):
                loaders_seen.append(sample['idxs'])
                yield sample
        _PlanLoader.__iter__ = tracking_iter
        run.step, run.eval_step = fake_step, fake_eval
        lrs = {}
        run.train()
        q.put((rank, config['ckpt_path'], trained, [lrs[e] for e in range(8)] + [run.optimizer.lr],
               (run.monitor_metric_best, float(run.scheduler.best), int(run.scheduler.num_bad_epochs))))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_entry_point_two_ranks_behave_like_one_run(tmp_path):
    import mrdis
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=200) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (_, path0, trained0, lrs0, best0), (_, path1, trained1, lrs1, best1) = res
    assert path0 == path1
    # sharding inside Run.train: 8 epochs x 9 rounds, disjoint within an epoch, the same number of steps on both ranks
    assert len(trained0) == len(trained1) == 8 * 9
    for e in range(8):
        a = [i for b in trained0[9 * e:9 * e + 9] for i in b]; b = [i for bb in trained1[9 * e:9 * e + 9] for i in bb]
        assert not set(a) & set(b) and len(set(a) | set(b)) == 36
    # the monitor every rank handed to the scheduler is the mean over the loader's 3 batches (constant), and the lr trajectory is
    # identical: six epochs without improvement -> one reduction by 0.1 (ReduceLROnPlateau(patience 5), main_missing.py:119)
    assert lrs0 == lrs1
    assert lrs0 == [2e-4] * 7 + [pytest.approx(2e-5)] * 2               # epochs 0..6 at the initial rate, reduced after the 7th val pass
    assert abs(best0[1] - 1.0) < 1e-12                                  # the monitor both schedulers saw: the constant mean
    assert best0 == best1
    # files: written once, by rank 0
    files = sorted(os.listdir(path0))
    assert files == sorted(['config.txt', 'config.yaml', 'model_best.pth.tar', 'stat.csv'] + [f'epoch{e:03d}.pth.tar' for e in range(8)])
    rows = open(os.path.join(path0, 'stat.csv')).read().strip().split('\n')
    assert len(rows) == 1 + 2 * 8                                       # header + (epoch row, val row) per epoch -- not doubled
    # epoch row 0 = mean over BOTH ranks' 18 iterations of the per-iteration values (sum of the batch's item indices + key offset)
    cols = rows[0].split(',')
    first = dict(zip(cols, rows[1].split(',')))
    want_all = np.mean([sum(b) for b in trained0[:9] + trained1[:9]]) + mrdis.LOSS_KEYS.index('all')
    assert abs(float(first['all']) - want_all) < 1e-6
    vcols = ['', 'info'] + sorted(list(mrdis.LOSS_KEYS) + ['rmse', 'psnr', 'ssim'])      # a val row carries the metrics too (util.py:854-866: no new header)
    val = dict(zip(vcols, rows[2].split(',')))
    assert abs(float(val['rmse']) - 1.0 / 3.0) < 1e-9 and abs(float(val['psnr']) - (20.0 * 4 + 21.0 * 2) / 6) < 1e-9   # image-weighted over ranks
    ck = torch.load(os.path.join(path0, 'epoch007.pth.tar'), weights_only=False)
    assert ck['epoch'] == 7 and abs(ck['monitor_metric'] - 1.0) < 1e-12 and set(ck) >= {'optimizer', 'scheduler', 'model', 'stat'}
