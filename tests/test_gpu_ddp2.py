"""Data-parallel rehearsal that fails where hardware would: TWO ranks (gloo, both on the one GPU of a test box) run the FULL TrainStep
(main_missing.py:165-289 on the HIP path) on RANK-DIVERGENT missing-modality batches -- rank 0's whole batch lacks modality 2, rank 1
lacks none -- under the reference's default accumulation schedule (accum = 2) with the adversarial second backward.  Rank 0's backward
prunes decoder 2's loss terms, so its gradient buckets complete in another order than rank 1's: the reducer must still pair its collectives,
OR the decoder gate flags across ranks and leave both ranks with the same weights.  Asserted: parameters and Adam moments identical across
ranks bit for bit, and identical to a single-process emulation that back-propagates both ranks' batches (each with its rank's own host RNG
streams), adds the two gradient arenas and applies the step with scale 1/2 -- the oracle of averaged gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

B, M, H, W, ITERS = 4, 3, 64, 96, 4


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _batch(mrdis, rank, it, dev):
    x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=50 + 10 * it + rank, drop=False)
    if rank == 0:                                        # modality 2 absent from rank 0's whole batch (util.py:538-542 zeroes the channels)
        mask[:, 2] = 0
        x[:, 7 * 2:7 * 3] = 0
    return x.to(dev).contiguous(memory_format=torch.channels_last), mask, mask_img.to(dev)


def _build(mrdis, dev, **kw):
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=8, lambda_adv_s=1.0)
    cfg = mrdis.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    return model, mrdis.TrainStep(model, cfg, ddp_buckets=4, **kw)


def _signature(model, step):
    """(sha256 of the bytes, sum of |values|) of the parameters and of both optimizers' moments: equal digests = bit-identical tensors"""
    import hashlib
    opt = step.optimizer
    out = []
    for t in (torch.cat([p.detach().reshape(-1) for p in model.parameters()]), opt.m, opt.v, opt.vmax, step.optimizer_d_s.m, step.optimizer_d_s.v):
        a = t.detach().cpu().contiguous().numpy()
        out.append((hashlib.sha256(a.tobytes()).hexdigest(), float(np.abs(a.astype(np.float64)).sum())))
    return out


def _emulate(mrdis, dev):
    """one process, both ranks' batches: gradients of rank 0 and rank 1 added (as the sum all-reduce does), step applied with scale 1 / 2"""
    model, step = _build(mrdis, dev)
    assert step.reducer is None and step.accum == 2              # (runs in the test process: no process group there)
    states = []
    for r in range(2):
        torch.manual_seed(100 + r); np.random.seed(100)
        states.append((torch.get_rng_state(), np.random.get_state()))
    opt, od = step.optimizer, step.optimizer_d_s
    for it in range(ITERS):
        do_step = step._advance(None)
        kept = []
        for r in range(2):
            torch.set_rng_state(states[r][0]); np.random.set_state(states[r][1])
            x, mask, mask_img = _batch(mrdis, r, it, dev)
            step._forward_backward(x, mask.to(dev), mask_img, mask, None, do_step, exchange=False)
            states[r] = (torch.get_rng_state(), np.random.get_state())
            kept.append((opt._g_full.clone(), od._g_full.clone()))
            opt._g_full.zero_(); od._g_full.zero_()
        opt._g_full.copy_(kept[0][0] + kept[1][0]); od._g_full.copy_(kept[0][1] + kept[1][1])
        step._apply(0.5, do_step)
    torch.cuda.synchronize()
    return _signature(model, step)


def _worker(rank, world, port, q, graph=False):
    import datetime
    import traceback
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    try:
        import mrdis
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        mrdis.hip.load()
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
        model, step = _build(mrdis, dev)
        red = step.reducer
        assert red is not None and red.world == world and red.exchanging
        if graph:                                                      # rank 1 (full masks) replays recorded graphs, rank 0 (a modality missing) cannot: eager
            step = mrdis.GraphedTrainStep(step, warm=1)
        torch.manual_seed(100 + rank); np.random.seed(100)             # eps per rank, the sim_s / adv_s pair identical on every rank
        losses = []
        for it in range(ITERS):
            x, mask, mask_img = _batch(mrdis, rank, it, dev)
            loss, _, _ = step(x, mask.to(dev), mask_img, mask)
            losses.append(float(loss))
        torch.cuda.synchronize()
        ex = red.exposed_ms()
        flags_seen = step.optimizer.gate_steps[:step.optimizer.n_flags].cpu().tolist()
        q.put((rank, 'ok', _signature(model, step), losses, dict(ex, graph_stats=dict(step.stats) if graph else None), flags_seen))
    except BaseException:                                            # noqa: BLE001 -- report instead of leaving the other rank in a collective
        q.put((rank, 'error', traceback.format_exc(), None, None, None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_with_divergent_missing_modalities_equal_the_averaged_gradient_oracle(mrdis):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert r[1] == 'ok', r[2]
    assert all(p.exitcode == 0 for p in procs)
    (_, _, sig0, losses0, ex0, flags0), (_, _, sig1, losses1, ex1, flags1) = res
    emu = _emulate(mrdis, torch.device('cuda:0'))
    names = ['weights', 'adam m', 'adam v', 'adam vmax', 'adam_d m', 'adam_d v']
    for n, a, b in zip(names, sig0, sig1):
        assert a == b, (n, 'differs across ranks', a, b)
    for n, a, e in zip(names, sig0, emu):
        assert a == e, (n, 'differs from the averaged-gradient oracle', a, e)
    assert losses0 != losses1                                        # the ranks really trained on different batches
    assert np.all(np.isfinite(losses0 + losses1))
    # every backward pass issued the same collectives on both ranks; decoder 2 stepped on both (its gate flag arrives from rank 1 only)
    assert ex0['finish_calls'] == ex1['finish_calls'] == ITERS + ITERS // 2 and ex0['bytes_reduced'] == ex1['bytes_reduced']
    assert flags0 == flags1 and all(f == ITERS // 2 for f in flags0), (flags0, flags1)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_graph_replay_on_one_eager_fallback_on_the_other(mrdis):
    """the same job with GraphedTrainStep: rank 1's batches (full masks) are replayed from recorded graphs, rank 0's (modality 2 missing: loss terms pruned)
    run eagerly -- both issue one all-reduce per gradient arena between backward and the optimizer, so they pair whatever each rank's mask decides.
    Weights and Adam moments identical across ranks and to the averaged-gradient oracle."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for r in res:
        assert r[1] == 'ok', r[2]
    (_, _, sig0, losses0, ex0, _), (_, _, sig1, losses1, ex1, _) = res
    assert ex0['graph_stats']['eager_irregular_mask'] == ITERS and ex0['graph_stats']['replays'] == 0, ex0
    assert ex1['graph_stats']['replays'] == 2 and ex1['graph_stats']['eager'] == 2, ex1
    emu = _emulate(mrdis, torch.device('cuda:0'))
    for n, a, b, e in zip(['weights', 'adam m', 'adam v', 'adam vmax', 'adam_d m', 'adam_d v'], sig0, sig1, emu):
        assert a == b, (n, 'differs across ranks', a, b)
        assert a == e, (n, 'differs from the averaged-gradient oracle', a, e)
