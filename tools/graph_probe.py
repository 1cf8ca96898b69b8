"""Feasibility probe for a graph-captured training step (round 6): warm up the eager step on a side stream, capture ONE step
(forward, losses, backward(s), clip, Adam) into a HIP graph with torch.cuda.graph, replay it, and time eager vs replay
(host time per step with an empty queue, and wall time per step back to back).  The captured step replays its capture-time host
draws (eps, pair selections): timing only, not a training mode."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
from mrdis import hip  # noqa: E402


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
    B, M, H, W = int(os.environ.get('PROBE_B', 32)), 4, int(os.environ.get('PROBE_HW', 256)), int(os.environ.get('PROBE_HW', 256))
    dev = torch.device('cuda:0')
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(input_height=H, input_width=W, batch_size=B, lambda_adv_s=1.0, compute_dtype=dtype)
    cfg = mrdis.derive_config(cfg, dev)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(cfg).train()
    step = mrdis.TrainStep(model, cfg)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=10)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last); maskd, mimgd = mask.to(dev), mask_img.to(dev)

    # mailbox without events (an event recorded inside a capture cannot be synchronised on later)
    def send(self, t_cpu, device):
        nbytes = t_cpu.numel() * t_cpu.element_size()
        k = self.next
        self.next = (k + 1) % self.NSLOT
        slot = self.buf[k * self.SLOT:k * self.SLOT + nbytes]
        slot.copy_(t_cpu.contiguous().view(-1).view(torch.uint8))
        out = torch.empty(t_cpu.shape, dtype=t_cpu.dtype, device=device)
        hip._chk(hip.load().mrdis_copy_bytes(slot.data_ptr(), hip._ptr(out), nbytes, hip._stream()), 'copy_bytes')
        return out
    hip._Mailbox.send = send

    def timed(fn, n):
        torch.cuda.synchronize()
        hosts = []
        t0 = time.perf_counter()
        for _ in range(n):
            h0 = time.perf_counter(); fn(); hosts.append((time.perf_counter() - h0) * 1e3)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(np.median(hosts))

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step(xd, maskd, mimgd, mask)
        torch.cuda.synchronize()
        ms_e, host_e = timed(lambda: step(xd, maskd, mimgd, mask), 5)
        print(f'[probe] eager on the side stream: {ms_e:.2f} ms/step, host {host_e:.2f} ms/step', flush=True)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    np.random.seed(5); torch.manual_seed(5)
    g = torch.cuda.CUDAGraph()
    hip.launch_counts(reset=True)
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        loss, parts, _ = step(xd, maskd, mimgd, mask)
    print(f'[probe] captured in {time.perf_counter() - t0:.2f} s; library launches in the graph: {hip.launch_counts()["all"]}', flush=True)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print(f'[probe] first replay ok, loss {float(loss):.5f}', flush=True)
    ms_g, host_g = timed(g.replay, 10)
    print(f'[probe] graph replay: {ms_g:.2f} ms/step, host {host_g:.2f} ms/step; loss {float(loss):.5f}', flush=True)
    # host time with an empty queue: replay, sync, replay ...
    hs = []
    for _ in range(5):
        torch.cuda.synchronize(); h0 = time.perf_counter(); g.replay(); hs.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    print(f'[probe] graph replay host time from an idle queue: {np.median(hs):.2f} ms; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)


if __name__ == '__main__':
    main()
