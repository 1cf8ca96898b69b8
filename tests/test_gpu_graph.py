"""The graph-replayed training step (trainer.GraphedTrainStep: main_missing.py:165-289 recorded once into HIP graphs, host draws
model.py:3159-3162 / :3485 re-drawn per replay through ops.host_value) against the eager TrainStep: same seeds, same batches with a
different missing-modality mask every iteration -> losses of every iteration and the weights after the last one BIT-IDENTICAL."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

DEV = torch.device('cuda:0')


def _batches(mrdis, n, B, M, H, W):
    out = []
    for seed in range(40, 400):                          # (B = 8, M = 4: about half of the drop-off masks leave every loss term in place)
        if len(out) == n:
            break
        x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=seed, drop=True)
        if mrdis.regular_mask(mask):                     # (a batch whose mask prunes a loss term is an eager step: tested separately)
            out.append((x.to(DEV).contiguous(memory_format=torch.channels_last), mask, mask_img.to(DEV)))
    assert len(out) == n
    return out


def _run(mrdis, graph, steps, batch_size, B, M=4, H=64, W=96, dtype='f32', force=False, warm=None):
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=batch_size, lambda_adv_s=1.0, compute_dtype=dtype)
    cfg = mrdis.derive_config(cfg, DEV)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.TrainStep(model, cfg, ddp_buckets=4, force_exchange=force)
    if graph:
        step = mrdis.GraphedTrainStep(step, warm=warm)
    data = _batches(mrdis, steps, B, M, H, W)
    torch.manual_seed(100); np.random.seed(100)
    losses, parts_all = [], []
    for x, mask, mask_img in data:
        loss, parts, _ = step(x, mask.to(DEV), mask_img, mask)
        losses.append(float(loss))                       # (reads the static buffer before the next replay overwrites it)
        parts_all.append({k: float(v) for k, v in parts.items()})
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    bufs = torch.cat([b.detach().float().reshape(-1) for b in model.buffers()]).cpu()
    mrdis.ops.set_compute_dtype('f32')
    return flat, bufs, losses, parts_all, step


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize('case', [(8, 16, 8, 'f32'), (12, 8, 8, 'f32'), (8, 16, 8, 'bf16')], ids=str)
def test_graph_replay_is_bit_identical_to_the_eager_step(mrdis, case):
    """(steps, config.batch_size, B, dtype): accum = 1 | the reference's default schedule (accum = 2: two recordings, accumulate / step) | bf16 storage"""
    steps, bs, B, dtype = case
    ref_w, ref_b, ref_l, ref_p, _ = _run(mrdis, False, steps, bs, B, dtype=dtype)
    got_w, got_b, got_l, got_p, step = _run(mrdis, True, steps, bs, B, dtype=dtype)
    st = step.stats
    nkeys = 1 if bs >= 16 else 2                         # (one recording per ordered adv_s pair and configuration: 12 for M = 4)
    assert st['captures'] == 12 * nkeys and st['eager'] == 2 * nkeys and st['replays'] == steps - 2 * nkeys, st
    assert got_l == ref_l, (got_l, ref_l)
    assert got_p == ref_p
    assert torch.equal(ref_w, got_w), float((ref_w - got_w).abs().max())
    assert torch.equal(ref_b, got_b)                     # BatchNorm running statistics and counters


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_irregular_mask_runs_eagerly_and_lr_change_records_again(mrdis):
    B, M, H, W = 8, 4, 64, 96
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=16, lambda_adv_s=1.0)
    cfg = mrdis.derive_config(cfg, DEV)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.GraphedTrainStep(mrdis.TrainStep(model, cfg), warm=1)
    (x, mask, mask_img), = _batches(mrdis, 1, B, M, H, W)
    for _ in range(3):
        step(x, mask.to(DEV), mask_img, mask)
    assert step.stats['captures'] == 12 and step.stats['replays'] == 2
    hole = mask.clone(); hole[:, 1] = 0                   # modality 1 absent from the whole batch: recon / mix terms are pruned
    assert not mrdis.regular_mask(hole)
    step(x, hole.to(DEV), mask_img, hole)
    assert step.stats['eager_irregular_mask'] == 1 and step.stats['captures'] == 12
    step.optimizer.param_groups[0]['lr'] *= 0.1           # ReduceLROnPlateau: the recorded Adam launch carries the old rate
    for _ in range(3):
        loss, _, _ = step(x, mask.to(DEV), mask_img, mask)
    assert step.stats['captures'] == 24 and len(step.entries) == 1 and np.isfinite(float(loss))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_graph_replay_with_the_exchange_between_two_graphs(mrdis):
    """data-parallel form on the one GPU of a test box: a one-rank RCCL group with force_exchange, so the step is recorded as two graphs
    (forward + backwards | clip + Adam) with the arenas all-reduced eagerly in between; a sum over one rank is the identity: bit-identical to the
    eager step without a process group."""
    assert not dist.is_initialized()
    ref_w, ref_b, ref_l, _, _ = _run(mrdis, False, 6, 8, 8)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1, device_id=DEV)
    try:
        got_w, got_b, got_l, _, step = _run(mrdis, True, 6, 8, 8, force=True, warm=1)
        assert step.reducer is not None and step.reducer.exchanging
        assert all(e['g2'] is not None for grp in step.entries.values() for e in grp.values()) and step.stats['captures'] == 24
        assert got_l == ref_l and torch.equal(ref_w, got_w) and torch.equal(ref_b, got_b)
        del step
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_a_step_that_cannot_be_recorded_trains_eagerly(mrdis):
    """a recording that raises half-way (here: the host-value block is made too small) must leave the step usable: the configuration falls back to eager
    iterations with the same results, the generators and the BatchNorm counters untouched by the aborted recording."""
    ref_w, ref_b, ref_l, _, _ = _run(mrdis, False, 4, 16, 8)
    cap = mrdis.ops.HostValues.CAP
    mrdis.ops.HostValues.CAP = 64
    try:
        with pytest.warns(UserWarning, match='recording failed'):
            got_w, got_b, got_l, _, step = _run(mrdis, True, 4, 16, 8, warm=1)
    finally:
        mrdis.ops.HostValues.CAP = cap
    assert step.stats['captures'] == 0 and step.stats['replays'] == 0 and step.stats['eager'] == 4
    assert got_l == ref_l and torch.equal(ref_w, got_w) and torch.equal(ref_b, got_b)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize('M', [3, 4])
def test_graph_replay_with_repeated_and_changing_pairs(mrdis, M):
    """the sim_s / adv_s pairs handed in by the wrapper: the SAME pair several iterations in a row, then other adv_s recordings in turn, against the eager
    step with the same pairs: losses, gradient norms and weights bit-identical.  (Found by the entry-point test: max_pool's backward zeroed dx with
    hipMemsetAsync, and that memset node did not run again on later replays of a recorded graph -- garbage gradients from the second replay on.  The
    kernel now writes every element of dx itself; no memset / memcpy call is left in the library.)"""
    ops = mrdis.ops
    pairs_seq = [((1, 2), (0, 1))] * 4 + [((0, 2), (2, 1)), ((0, 2), (2, 1)), ((2, 0), (1, 0)), ((0, 1), (0, 2)), ((1, 0), (1, 2)), ((1, 2), (2, 0))]
    res = {}
    for graph in (False, True):
        cfg = dict(mrdis.DEFAULT_CONFIG)
        cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=64, input_width=96, batch_size=16, lambda_adv_s=1.0)
        cfg = mrdis.derive_config(cfg, DEV)
        torch.manual_seed(10); np.random.seed(10)
        model = mrdis.build_model(cfg).train()
        base = mrdis.TrainStep(model, cfg)
        step = mrdis.GraphedTrainStep(base, warm=1) if graph else base
        torch.manual_seed(100); np.random.seed(100)
        out = []
        for k, (ps, pa) in enumerate(pairs_seq):
            x, mask, mask_img = mrdis.synthetic_batch(8, M, 64, 96, seed=60 + k)
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
            pairs = {'sim_s': ps, 'adv_s': pa}
            if graph:
                step._predraw = lambda p=pairs: dict(p)
                loss, _, _ = step(xd, mask.to(DEV), mask_img.to(DEV), mask)
            else:
                ops.set_forced_pairs(pairs)
                try:
                    loss, _, _ = step(xd, mask.to(DEV), mask_img.to(DEV), mask)
                finally:
                    ops.set_forced_pairs(None)
            out.append((float(loss), float(base.optimizer.flat_p.double().abs().sum()), float(base.last_grad_norm_sq[0])))
        res[graph] = (out, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
    assert res[False][0] == res[True][0], [a == b for a, b in zip(res[False][0], res[True][0])]
    assert torch.equal(res[False][1], res[True][1])
