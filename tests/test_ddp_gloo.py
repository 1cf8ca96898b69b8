"""N > 1 path on CPU: world_size-2 gloo run of the gradient exchange (bucketed, hook-driven
all-reduce over the flat gradient arena).  The HIP kernels are not involved: the arena and the
reducer are plain torch plumbing, which is exactly what this exercises."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(7, 13)          # sizes that are not multiples of 4: exercises arena padding
        self.b = nn.Linear(13, 5)
        self.c = nn.Linear(5, 3)
        self.unused = nn.Linear(3, 3)      # never receives a gradient (like output_decoder)

    def forward(self, x):
        h = torch.tanh(self.a(x))
        return self.c(torch.tanh(self.b(h))) + self.c(torch.tanh(self.b(-h)))   # shared weights, used twice


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mrdis
        torch.manual_seed(0)
        model = Tiny()
        opt = mrdis.ArenaAdam(model.parameters(), lr=1e-3)
        red = mrdis.GradAllReduce(opt, buckets=3)
        errs = []
        for step in range(3):
            g = torch.Generator().manual_seed(100 * step + rank)
            x = torch.randn(4, 7, generator=g)
            # expected: mean over ranks of the local gradient
            local = torch.autograd.grad(model(x).pow(2).sum(), [p for n, p in model.named_parameters() if 'unused' not in n])
            expect = []
            for t in local:
                t = t.clone(); dist.all_reduce(t); expect.append(t / world)
            opt.zero_grad()
            red.begin()
            model(x).pow(2).sum().backward()
            scale = red.finish()
            if opt.used is None:
                opt._build()
            got = [p.grad * scale for n, p in model.named_parameters() if 'unused' not in n]
            for a, b in zip(got, expect):
                errs.append(float((a - b).abs().max()))
            assert model.unused.weight.grad is None
            assert all(p.grad.data_ptr() >= opt.flat_g.data_ptr() for p in opt.used)    # views into the arena
        q.put((rank, max(errs), opt.numel, len(opt.used)))
    finally:
        dist.destroy_process_group()


def _worker_two_optimizers(rank, world, port, q):
    """The data-parallel plumbing of TrainStep with the adversarial second backward and accum == 2, on a CPU model:
    static arena (built before the first backward, identical on every rank even when a rank's batch leaves a parameter
    without gradient), a second optimizer sharing the weights with its own gradient buffer (p.grad re-pointed, no clones),
    both reduced through the same bucketed reducer, gate flags OR-ed across ranks by the same sum all-reduce, and the
    accumulation rule (each micro-batch gradient reduced exactly once, then added to the accumulator)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mrdis
        torch.manual_seed(0)
        model = Tiny()
        used = [p for n, p in model.named_parameters() if 'unused' not in n]
        opt = mrdis.ArenaAdam(model.parameters(), lr=1e-3, used=used)
        opt.set_gates([[model.c.weight, model.c.bias], [model.a.weight]])
        od = mrdis.ArenaAdam(model.parameters(), lr=1e-3, weight_decay=0.0, share_weights_of=opt)
        red = mrdis.GradAllReduce(opt, buckets=3)
        assert od.flat_p.data_ptr() == opt.flat_p.data_ptr() and od.flat_g.data_ptr() != opt.flat_g.data_ptr()
        errs = []
        pos = {id(p): k for k, p in enumerate(opt.used)}
        acc = torch.zeros_like(opt._g_full)
        acc_ref = [torch.zeros_like(p) for p in used]
        for it in range(4):
            g = torch.Generator().manual_seed(100 * it + rank)
            x = torch.randn(4, 7, generator=g)

            def losses():
                y = model(x)
                return y.pow(2).sum(), (y[:, :1] - 1).abs().sum()      # "generator" loss, "discriminator" loss (reaches a, b, c too)
            # expected: mean over ranks of the local gradients of both losses
            lg, ld = losses()
            eg = list(torch.autograd.grad(lg, used, retain_graph=True)); ed = list(torch.autograd.grad(ld, used))
            for t in eg + ed:
                dist.all_reduce(t); t /= world
            # --- the TrainStep sequence
            lg, ld = losses()
            flags = torch.tensor([1.0 if rank == it % 2 else 0.0, 0.0])           # group 0 active on one rank only, group 1 on none
            opt.mark_active(flags)
            red.begin(opt); lg.backward(retain_graph=True); scale = red.finish()
            od.attach_grads()
            red.begin(od); ld.backward(); red.finish()
            opt.attach_grads()
            for p, a, b in zip(used, eg, ed):
                k = pos[id(p)]
                errs.append(float((opt.grad_views[k] * scale - a).abs().max()))
                errs.append(float((od.grad_views[k] * scale - b).abs().max()))
            assert list(opt.gate_flags) == [1.0, 0.0]                          # OR over the ranks, by the same all-reduce
            assert model.unused.weight.grad is None
            # accumulation (accum == 2): reduced micro-batch gradient joins the accumulator once; clip acts on the accumulator
            acc.add_(opt._g_full, alpha=scale); opt.zero_grad(); od.zero_grad()
            n = opt.numel
            coef = torch.clamp(1.0 / (acc[:n].norm() + 1e-6), max=1.0); acc[:n].mul_(coef)
            for r, a in zip(acc_ref, eg):
                r.add_(a)
            tn = torch.sqrt(sum((r ** 2).sum() for r in acc_ref)); c = torch.clamp(1.0 / (tn + 1e-6), max=1.0)
            for r in acc_ref:
                r.mul_(c)
            for p, r in zip(used, acc_ref):
                o = opt.offsets[pos[id(p)]]
                errs.append(float((acc[o:o + p.numel()].view(p.shape) - r).abs().max()))
            if it % 2 == 1:
                acc.zero_()
                for r in acc_ref:
                    r.zero_()
        q.put((rank, max(errs), opt.numel, len(opt.used)))
    finally:
        dist.destroy_process_group()


class _SinkLinearFn(torch.autograd.Function):
    """y = x W^T + b whose parameter gradients are ADDED to p.grad inside backward and never handed to autograd -- the CPU twin of
    the in-kernel gradient sinks of the HIP path (ops._grad_sink): no post-accumulate hook fires for W and b."""

    @staticmethod
    def forward(ctx, x, lin):
        ctx.lin = lin
        ctx.save_for_backward(x)
        return x @ lin.weight.detach().t() + lin.bias.detach()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lin = ctx.lin
        lin.weight.grad.add_(dy.t() @ x); lin.bias.grad.add_(dy.sum(0))
        return dy @ lin.weight.detach(), None


def _worker_sinks(rank, world, port, q):
    """Parameters whose gradients are in-kernel sinks: no hook fires, their buckets leave when the group's backward node reports
    them (GradAllReduce.mark_ready, called by ops._MixAllLayers.backward on the GPU path) -- here from a tensor hook that runs
    mid-backward -- or from finish().  Arena laid out in completion order (ArenaAdam(order=...)), buckets cut at group edges."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import mrdis
        torch.manual_seed(0)
        model = Tiny()
        sink = [model.b.weight, model.b.bias, model.c.weight, model.c.bias]
        for p in sink:
            p._mrdis_sink = True
        used = [p for n, p in model.named_parameters() if 'unused' not in n]
        order = [[model.c.weight, model.c.bias], [model.b.weight, model.b.bias], [model.a.weight, model.a.bias]]
        opt = mrdis.ArenaAdam(model.parameters(), lr=1e-3, used=used, order=order)
        assert [id(p) for p in opt.used] == [id(p) for g in order for p in g]          # completion order = arena order
        assert opt.group_edges == [0, opt.offsets[2], opt.offsets[4], opt.numel]
        od = mrdis.ArenaAdam(model.parameters(), lr=1e-3, weight_decay=0.0, share_weights_of=opt)     # optimizer_d_s: the same layout, its own gradients
        assert [id(p) for p in od.used] == [id(p) for p in opt.used] and od.offsets == opt.offsets and od.group_edges == opt.group_edges
        assert all(gv.shape == p.shape for gv, p in zip(od.grad_views, od.used))
        red = mrdis.GradAllReduce(opt)
        errs, early = [], []
        for it in range(3):
            g = torch.Generator().manual_seed(100 * it + rank)
            x = torch.randn(4, 7, generator=g)
            ref = Tiny(); ref.load_state_dict(model.state_dict())
            expect = list(torch.autograd.grad(ref(x).pow(2).sum(), [p for n, p in ref.named_parameters() if 'unused' not in n]))
            for t in expect:
                dist.all_reduce(t); t /= world
            expect = {n: t for (n, _), t in zip([(n, p) for n, p in ref.named_parameters() if 'unused' not in n], expect)}
            opt.zero_grad()
            h = torch.tanh(model.a(x))
            if it < 2:
                # what the mixing group's backward node does: by the time dL/dh exists, b's and c's sink gradients are complete
                h.register_hook(lambda g_, r=red: r.mark_ready(sink))
            y = _SinkLinearFn.apply(torch.tanh(_SinkLinearFn.apply(h, model.b)), model.c) + \
                _SinkLinearFn.apply(torch.tanh(_SinkLinearFn.apply(-h, model.b)), model.c)
            red.begin()
            before = red.early_buckets
            y.pow(2).sum().backward()
            scale = red.finish()
            early.append(red.early_buckets - before)
            for n, p in model.named_parameters():
                if 'unused' not in n:
                    errs.append(float((p.grad * scale - expect[n]).abs().max()))
        # iterations 0, 1: the c and b buckets left from mark_ready, a's from its own hooks; iteration 2 (nobody reports the
        # sinks): a's buckets are complete during backward but wait for the sink buckets in front of them (buckets leave in
        # index order, the same on every rank), which go from finish()
        q.put((rank, max(errs), early, red.nbuckets))
    finally:
        dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    return sorted(q.get(timeout=5) for _ in range(2))


@pytest.mark.timeout(180)
def test_two_optimizers_accumulation_world2():
    for rank, err, numel, nused in _run(_worker_two_optimizers):
        assert err < 1e-6, (rank, err)
        assert nused == 6


@pytest.mark.timeout(180)
def test_sink_parameters_leave_from_mark_ready_or_finish_world2():
    for rank, err, early, nb in _run(_worker_sinks):
        assert err < 1e-6, (rank, err)
        # groups c | b | a at 20 | 76 | 108 floats: the two beyond a third of the arena are cut in two -> 5 buckets
        assert nb == 5 and early == [5, 5, 0], (early, nb)


@pytest.mark.timeout(180)
def test_bucketed_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    for rank, err, numel, nused in res:
        assert err < 1e-6, (rank, err)
        assert nused == 6 and numel % 4 == 0
