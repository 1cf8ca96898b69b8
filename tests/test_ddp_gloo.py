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
