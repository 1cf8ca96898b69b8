"""The gradient exchange over RCCL on the one GPU a test box has: a process group of ONE rank (backend nccl, device_id = cuda:0)
with `force_exchange`, which bypasses the world == 1 early-outs of GradAllReduce, so every collective of a data-parallel step
is really issued: `all_reduce(async_op=True)` on slices of the gradient arena from the autograd engine's thread (mark_ready from
the mixing groups' backward nodes, post-accumulate hooks), the waits on the compute stream in finish(), the gate-flag tail, the
second (discriminator-loss) backward into the second optimizer's buffer.  A sum over one rank is the identity: the weights must
equal a run without any process group BIT FOR BIT.  Replaces main_missing.py:268-284 (backward + step) under data parallelism."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _run(mrdis, dev, steps, batch_size, compute_dtype, force):
    B, M, H, W = 4, 3, 64, 96
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=batch_size, lambda_adv_s=1.0,
               compute_dtype=compute_dtype)
    cfg = mrdis.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.TrainStep(model, cfg, ddp_buckets=4, force_exchange=force)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=3, drop=True)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    torch.manual_seed(100); np.random.seed(100)
    losses = []
    for _ in range(steps):
        loss, _, _ = step(xd, mask.to(dev), mask_img.to(dev), mask)
        losses.append(float(loss))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    mrdis.ops.set_compute_dtype('f32')
    return flat, losses, step


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_single_rank_rccl_group_runs_every_collective_and_changes_nothing(mrdis):
    dev = torch.device('cuda:0')
    assert not dist.is_initialized()
    # (steps, config.batch_size, compute_dtype): accum = 1 | the reference's default schedule (accum = 2) | bf16 storage
    cases = [(2, 16, 'f32'), (4, 8, 'f32'), (2, 16, 'bf16')]
    refs = [_run(mrdis, dev, st, bs, cd, False) for st, bs, cd in cases]
    assert all(r[2].reducer is None for r in refs)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == 'nccl'
        for (st, bs, cd), (ref, ref_losses, _) in zip(cases, refs):
            got, losses, step = _run(mrdis, dev, st, bs, cd, True)
            red = step.reducer
            assert red is not None and red.world == 1 and red.exchanging
            assert torch.equal(ref, got), (cd, bs, float((ref - got).abs().max()))
            assert losses == ref_losses
            ex = red.exposed_ms()
            # two backward passes per optimizer step (generator loss, discriminator loss on stepping iterations)
            assert ex['finish_calls'] >= st and ex['buckets'] >= 4
            assert ex['early_buckets'] > 0                               # buckets that left DURING backward, from the autograd thread
            assert ex['bytes_reduced'] >= 4 * step.optimizer.numel * st
            assert np.isfinite(ex['exposed_ms'])
        # without force_exchange a one-rank group stays silent (what the driver's N = 1 bench line runs)
        _, _, quiet = _run(mrdis, dev, 1, 16, 'f32', False)
        assert quiet.reducer is not None and not quiet.reducer.exchanging and quiet.reducer.exposed_ms()['bytes_reduced'] == 0
        # timing diagnostics: the event pair around the waits on the compute stream
        _, _, timed = _run(mrdis, dev, 1, 16, 'f32', True)
        timed.reducer.exposed_ms()                                      # reset the counters of the step _run took
        timed.reducer.timing = True
        x, mask, mask_img = mrdis.synthetic_batch(4, 3, 64, 96, seed=3, drop=True)
        timed(x.to(dev).contiguous(memory_format=torch.channels_last), mask.to(dev), mask_img.to(dev), mask)
        ex = timed.reducer.exposed_ms()
        assert ex['exposed_ms'] >= 0.0 and np.isfinite(ex['exposed_ms']) and ex['finish_calls'] == 2
    finally:
        dist.destroy_process_group()
    assert not dist.is_initialized()
