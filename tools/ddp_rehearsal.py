"""Rehearsal of the data-parallel TrainStep on the real HIP path with world_size > 1 on ONE GPU box: every rank uses
cuda:0 and the exchange runs over gloo (RCCL needs one device per rank).  Every rank feeds the SAME batch with the same
seeds, so the averaged gradients equal the single-process ones bit for bit ((g + g) / 2 == g in fp32) and, the kernels
being bit-reproducible, the weights after k steps must be IDENTICAL to a single-process run and across ranks.  That
pins the parts the CPU gloo test cannot reach: gradient sinks written inside backward kernels, hook-driven bucket
launches racing the rest of the backward pass, the adversarial second backward into the second optimizer's buffer, gradient
accumulation (the reference's default schedule), gated decoders, and the bf16 mode.

Second part (round 4): the entry point itself (`entry_point` below) -- Run.train over two ranks that feed DIFFERENT shards.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_rehearsal.py
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402


def run(steps, B, M, H, W, dev, batch_size=16, compute_dtype='f32'):
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=[f'm{i}' for i in range(M)], input_height=H, input_width=W, batch_size=batch_size, lambda_adv_s=1.0,
               compute_dtype=compute_dtype)
    cfg = mrdis.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.TrainStep(model, cfg, ddp_buckets=4)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=3, drop=True)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    torch.manual_seed(100); np.random.seed(100)
    losses = []
    for _ in range(steps):
        loss, parts, _ = step(xd, mask.to(dev), mask_img.to(dev), mask)
        losses.append(float(loss))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    return flat, losses, step


def entry_point(rank, world, root):
    """The ENTRY POINT as the data-parallel job (mrdis/train.py = main_missing.py): two epochs of Run.train with the ranks feeding
    DIFFERENT shards of one synthetic fold (rank-sharded BatchLoader, drop-off masks, adversarial second backward), validation
    and ReduceLROnPlateau on the all-reduced monitor, checkpoints by rank 0.  Asserts: the same number of steps on every rank,
    weights + Adam moments + lr identical across ranks after training (the exchange is the ONLY thing that couples them),
    every file written once."""
    T = mrdis.train
    over = dict(contrast_list=['T1', 'T1c', 'T2'], input_height=64, input_width=96, batch_size=4, epochs=2, gpu='0', dropoff=True,
                lambda_adv_s=1.0, data_source='synthetic', ckpt_root=os.path.join(root, 'ckpt'), ckpt_timelabel=None)
    cfgfile = os.path.join(root, 'config.yaml')
    if rank == 0:
        import yaml
        os.makedirs(root, exist_ok=True)
        with open(cfgfile, 'w') as f:
            yaml.dump(over, f)
    dist.barrier()
    config = T.setup_config(cfgfile)
    lines = []
    run = T.Run(config, log=lambda m: lines.append(m))
    steps = []
    inner = run.step

    def counting_step(*a, **k):
        steps.append(1)
        return inner(*a, **k)
    counting_step.optimizer, counting_step.optimizer_d_s = inner.optimizer, inner.optimizer_d_s
    run.step = counting_step
    run.train()
    torch.cuda.synchronize()
    opt = run.optimizer
    sig = [float(torch.cat([p.detach().reshape(-1) for p in run.model.parameters()]).double().abs().sum()),
           float(opt.m.double().abs().sum()), float(opt.v.double().abs().sum()), float(opt.lr), len(steps)]
    bn = float(sum(b.double().abs().sum() for n, b in run.model.named_buffers() if 'running' in n))
    sigs = [None] * world
    dist.all_gather_object(sigs, (sig, bn))
    same = all(s[0] == sigs[0][0] for s in sigs)
    ex = inner.reducer.exposed_ms()
    ok = same
    if rank == 0:
        files = sorted(os.listdir(config['ckpt_path']))
        want = sorted(['config.txt', 'config.yaml', 'stat.csv', 'model_best.pth.tar', 'epoch000.pth.tar', 'epoch001.pth.tar'])
        rows = open(os.path.join(config['ckpt_path'], 'stat.csv')).read().strip().split('\n')
        ok = ok and files == want and len(rows) == 1 + 2 * 2
        print(f'[ddp rehearsal] entry point (Run.train, 2 epochs, world {world}, ranks on different shards): {len(steps)} steps per rank; weights / Adam m / v / lr '
              f'identical across ranks: {same} ({[s[0] for s in sigs]}); BatchNorm running statistics per replica (sums {[round(s[1], 4) for s in sigs]}); '
              f'{ex["early_buckets"]} of {ex["buckets"] * ex["finish_calls"]} bucket launches during backward; files {files} (written once: {files == want}); '
              f'stat.csv rows {len(rows)}', flush=True)
        for m in lines:
            if m.startswith('epoch '):
                print('[ddp rehearsal]   ' + m, flush=True)
    return ok


def main():
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    mrdis.hip.load()
    B, M, H, W = 4, 3, 64, 96
    # (steps, config.batch_size, compute_dtype): accum = 1; the reference's default schedule (accum = 2, two optimizer steps); bf16 storage
    cases = [(3, 16, 'f32'), (4, 8, 'f32'), (2, 16, 'bf16')]
    refs = [run(st, B, M, H, W, dev, bs, cd)[:2] for st, bs, cd in cases]        # single process: no reducer
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ok = True
    for (st, bs, cd), (ref, ref_losses) in zip(cases, refs):
        got, losses, step = run(st, B, M, H, W, dev, bs, cd)
        assert step.reducer is not None and step.reducer.world == world
        same = bool(torch.equal(ref, got))
        maxdiff = float((ref - got).abs().max())
        sums = [None] * world
        dist.all_gather_object(sums, (float(got.double().sum()), float(got.double().abs().sum())))
        across = all(s == sums[0] for s in sums)
        ex = step.reducer.exposed_ms()
        if rank == 0:
            print(f'[ddp rehearsal]   exchange: {ex["buckets"]} buckets cut at the completion-group edges, {ex["early_buckets"]} of '
                  f'{ex["buckets"] * ex["finish_calls"]} bucket launches left DURING backward (mark_ready / hooks), {ex["finish_calls"]} backward passes, '
                  f'{ex["bytes_reduced"] / 1e6:.1f} MB reduced', flush=True)
            print(f'[ddp rehearsal] world {world}, batch_size {bs} (accum {step.accum}), compute_dtype {cd}: weights after {st} iterations identical '
                  f'to single-process: {same} (max |diff| {maxdiff:.3e}); identical across ranks: {across}; losses {losses} vs {ref_losses}', flush=True)
        ok = ok and same and across
    import tempfile
    root = [tempfile.mkdtemp(prefix='mrdis_ddp_') if rank == 0 else None]
    dist.broadcast_object_list(root, src=0)
    ok = entry_point(rank, world, root[0]) and ok
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    dist.barrier()
    dist.destroy_process_group()
    if not all(flags):
        sys.exit(1)


if __name__ == '__main__':
    main()
