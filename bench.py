"""bench.py -- headline benchmark of the MI355X-native hot path.

    python bench.py [--gpus N --steps K --warmup W]           # N = 1; N > 1 without a launcher: starts the N ranks itself (child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W  # N > 1, one rank per GPU (RCCL)

Metric (BASELINE.json): MR slices/sec of one full training step -- anatomy + modality
encoders, 4 + 12 SPADE decodes, recon_x + recon_x_mix + latent_z cycle + sim_s + sim_z +
adversarial losses, backward, clip, Adam(amsgrad) -- a "slice" being one batch element with
all M modalities.  Workload at N = 1 is BASELINE.json configs[1]: 4-modality 240x240 fp32
slices, batch 32 per GPU; 240 is not a multiple of 32 (the U-Net needs five stride-2 stages,
reference model.py:2192), so each slice is padded with background to 256x256 -- no pixel of
the named workload is dropped (the reference's own 160x192 crop is available with
--fit crop).  Weak scaling: per-GPU batch fixed, gradients mean-all-reduced over RCCL.
Inputs are synthetic BraTS-shaped slices already resident in HBM; weights random-init.

One JSON line is printed by rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3       # dense f32-input MFMA peak
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16 MFMA peak (MI355X_MICROARCH.md; never the 2:1-sparsity figure)
# north-star conv x(32,4,240,240) -> y(32,32,240,240), W(32,4,3,3), fp32
NS = dict(N=32, Ci=4, Co=32, H=240, W=240, k=3)
NS_BYTES = 4 * (NS['N'] * NS['Ci'] * NS['H'] * NS['W'] + NS['N'] * NS['Co'] * NS['H'] * NS['W'] + NS['Co'] * NS['Ci'] * 9 + NS['Co'])
# fwd+bwd conv FLOPs per slice at 160x192, M = 4 (torch FlopCounter on the reference)
FLOP_PER_SLICE_160x192 = 2.784e11


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1, help='ranks = GPUs of this node; must equal WORLD_SIZE when started by a launcher (mandatory there: the default 1 is refused '
                                                        'under WORLD_SIZE > 1); N > 1 without a launcher starts the N ranks itself')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (slices)')
    ap.add_argument('--modalities', type=int, default=4)
    ap.add_argument('--slice', type=int, default=240, help='synthetic slice extent (BraTS: 240)')
    ap.add_argument('--fit', choices=['pad', 'crop'], default='pad', help='240 -> 256x256 pad (default) or 160x192 crop')
    ap.add_argument('--no-adv', action='store_true', help='lambda_adv_s = 0 (the shipped config.yaml)')
    ap.add_argument('--drop', action='store_true', help='missing-modality batches (BASELINE configs[3])')
    ap.add_argument('--recon-y', action='store_true', help="lambda_recon_y = 1: adds the 'U+SA' output decoder + segmentation loss (not the headline config)")
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16', 'bf16m'],
                    help="f32 (headline: exact fp32 MFMA, fp32 activations) | bf16 (BASELINE configs[2]: bf16 activations in HBM, bf16 MFMA operands, "
                         "fp32 accumulate / statistics / master weights / optimizer) | bf16m (bf16 MFMA operands on fp32 activations)")
    ap.add_argument('--graph', action='store_true', help='the timed steps as HIP-graph replays (trainer.GraphedTrainStep; also MRDIS_GRAPH=1): forward, losses, backward, clip, Adam '
                                                         'recorded once after two eager iterations; >= 3 untimed warm-up steps are run so that the timed region only replays')
    ap.add_argument('--no-graph-leg', action='store_true', help='skip the extra graph-replay timing beside the (eager) headline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (the stored number is reported instead)')
    ap.add_argument('--no-direct', action='store_true', help='skip the extra direct-kernels-only (wino = 0) timing')
    ap.add_argument('--dry-run', action='store_true', help='launcher rehearsal without a GPU: every rank joins a gloo group and reports in; rank 0 prints the '
                                                           'ranks it heard from (tests/test_bench_launch.py)')
    ap.add_argument('--roofline-only', action='store_true', help='only the roofline legs (north-star conv + the dominant kernels of the step), '
                                                                 'no training step: what tools/profile_round.sh profiles separately from the step')
    return ap.parse_args()


def _time_conv(hip, xs, w, b, ys, iters, warm_ms=0.0):
    """average launch duration (us) over `iters` launches cycling through the (x, y) buffer pairs; HIP events on the launch
    stream (torch's current stream is the stream the C ABI is handed).  warm_ms = 0: the protocol of rounds 1-5 (a few warm-up launches, then `iters`
    timed ones: a BURST of ~1 ms from a lightly loaded chip); warm_ms > 0: back-to-back launches for that long first -- the SUSTAINED rate, which for
    this store-bound kernel is 10-25 % lower (rocprofv3 over 335 consecutive launches: 42.9 us min, 56.6 avg, 71.5 max; round 6)."""
    n = len(xs)
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    for i in range(max(3, n)):
        hip.conv2d_fwd(xs[i % n], w, b, 3, 3, 1, 1, out=ys[i % n])
    w1.record(); torch.cuda.synchronize()
    if warm_ms > 0:
        for i in range(max(0, int(max(3, n) * (warm_ms / max(w0.elapsed_time(w1), 0.05) - 1.0)))):
            hip.conv2d_fwd(xs[i % n], w, b, 3, 3, 1, 1, out=ys[i % n])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        hip.conv2d_fwd(xs[i % n], w, b, 3, 3, 1, 1, out=ys[i % n])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def _run_group(cmd, cwd, env, timeout_s):
    """run a child in its OWN process group and, on timeout, kill the whole group (rocprofv3 is a wrapper: killing only it would orphan the python
    grandchild that holds the GPU) -- then wait, so nothing outlives this call.  Returns (returncode or None on timeout, combined output)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = p.communicate(timeout=timeout_s)
        return p.returncode, out
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, _ = p.communicate()
        return None, out


def measure_traffic_inrun(timeout_s=240, script=('northstar_conv.py', '12'), kernel_substr='c4conv', n_last=12, what='tools/northstar_conv.py'):
    """HBM bytes per launch of ONE kernel from the PMC counters, measured IN THIS RUN: two child processes (never an exec: this process has
    initialised the GPU), `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/<script>` and the same with WRITE_SIZE (separate passes, the
    program itself after `--`), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 on gfx950, both in KiB).  Default: the north-star conv;
    roofline_step uses it for the full-resolution F(4x4) kernels.  Returns (bytes, provenance, kernel name seen in the counter file), or
    (None, reason, None) when rocprofv3 is missing / a pass fails (the caller then reports the stored number and says so)."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found', None
    if os.environ.get('MRDIS_BENCH_NO_PMC'):
        return None, 'MRDIS_BENCH_NO_PMC set', None
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    got = {}
    kname = None
    t0 = time.time()
    with tempfile.TemporaryDirectory(prefix='mrdis_pmc_', dir='/tmp') as d:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', counter.lower(), '--',
                   sys.executable if os.path.basename(sys.executable).startswith('python') else 'python3', os.path.join(ROOT, 'tools', script[0])] + list(script[1:])
            try:
                rc, txt = _run_group(cmd, '/tmp', env, timeout_s)
            except Exception as ex:                                   # noqa: BLE001 -- a profiler that cannot run must not take the benchmark down
                return None, f'{counter} pass: {type(ex).__name__}', None
            if rc is None:
                return None, f'{counter} pass: timed out after {timeout_s} s (process group killed)', None
            if rc != 0:
                return None, f'{counter} pass: rc {rc}: ' + (txt or '')[-160:].replace('\n', ' '), None
            path = None
            for root_, _, files in os.walk(d):
                for f in files:
                    if f.startswith(counter.lower()) and f.endswith('counter_collection.csv'):
                        path = os.path.join(root_, f)
            if path is None:
                return None, f'{counter} pass: no counter_collection.csv', None
            rows = [row for row in csv.DictReader(open(path)) if row['Counter_Name'] == counter and kernel_substr in row['Kernel_Name']]
            vals = [float(row['Counter_Value']) for row in rows]
            if len(vals) < min(8, n_last):
                return None, f'{counter} pass: {len(vals)} launches of {kernel_substr} in the counter file', None
            kname = rows[-1]['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            vals = vals[-n_last:]                                     # the timed launches (the check / warm-up launches come first)
            got[counter] = sum(vals) / len(vals)
    rd, wr = 2.0 * got['FETCH_SIZE'] * 1024.0, got['WRITE_SIZE'] * 1024.0
    return rd + wr, (f'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate child passes of {what}, '
                     f'{time.time() - t0:.0f} s), mean over {n_last} launches; read = 2 x FETCH_SIZE x 1024 = {rd / 1e6:.1f} MB (gfx950 half-count correction), '
                     f'write = WRITE_SIZE x 1024 = {wr / 1e6:.1f} MB'), kname


def roofline_conv(mrdis, dev, iters=24, extras=True, pmc_inrun=False):
    """The north-star 3x3 conv forward.  Its output is checked against torch fp32 on the host first.  Timed
    three ways: `achieved` = ROTATING buffers (4 x/y pairs, 1.06 GB > the 256 MiB Infinity Cache, so every launch reads and
    writes HBM), `single_buffer` = one x/y pair re-used (the 236 MB output partly lives in the Infinity Cache between
    launches), `in_step_256` = the same layer at the size it has inside the timed training step (256x256 padded, 302 MB)."""
    import torch.nn.functional as F
    hip = mrdis.hip
    g = torch.Generator().manual_seed(1)
    x_cpu = torch.randn(NS['N'], NS['Ci'], NS['H'], NS['W'], generator=g)
    w_cpu = torch.randn(NS['Co'], NS['Ci'], 3, 3, generator=g) * 0.1
    b_cpu = torch.randn(NS['Co'], generator=g) * 0.1
    w = w_cpu.permute(2, 3, 1, 0).reshape(9, NS['Ci'], NS['Co']).contiguous().to(dev)
    b = b_cpu.to(dev)
    nrot = 4
    xs = [x_cpu.roll(i, 0).to(dev).contiguous(memory_format=torch.channels_last) for i in range(nrot)]
    ys = [hip.empty_nhwc(NS['N'], NS['Co'], NS['H'], NS['W'], dev) for _ in range(nrot)]
    hip.conv2d_fwd(xs[0], w, b, 3, 3, 1, 1, out=ys[0])
    want = F.conv2d(x_cpu, w_cpu, b_cpu, 1, 1)
    err = float((ys[0].cpu() - want).abs().max()) / float(want.abs().max())
    assert err <= 1e-5, f'north-star conv output differs from torch fp32: rel {err:.2e}'
    hip.launch_counts(reset=True)
    us_rot = _time_conv(hip, xs, w, b, ys, iters)
    six = hip.launch_counts()['split6_c4'] > 0               # which form the library's policy launched (option split6)
    if not extras:          # tools/northstar_conv.py under rocprofv3 --pmc: only launches of the north-star shape
        return {'us_per_launch': round(us_rot, 2), 'achieved': round(NS_BYTES / (us_rot * 1e-6) / 1e9, 1)}
    us_f32 = None
    if six:                 # the exact-fp32 MFMA form of the same kernel (option split6 = 0), same buffers, same run, same burst protocol
        with hip.option('split6', 0):
            us_f32 = _time_conv(hip, xs, w, b, ys, iters)
    us_one = _time_conv(hip, xs[:1], w, b, ys[:1], iters)
    # the ceiling of a store-only kernel on THIS device, over the same rotating output buffers (236 MB each): the layer's algorithmic traffic is 89 % stores
    for i in range(4):
        hip.stream_fill(ys[i % nrot])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        hip.stream_fill(ys[i % nrot])
    e1.record()
    torch.cuda.synchronize()
    us_fill = e0.elapsed_time(e1) * 1e3 / iters
    fill_gbs = ys[0].numel() * 4 / (us_fill * 1e-6) / 1e9
    n6 = NS['N'] * 256 * 256
    bytes6 = 4 * (n6 * NS['Ci'] + n6 * NS['Co'] + NS['Co'] * NS['Ci'] * 9 + NS['Co'])
    xs6 = [torch.randn(NS['N'], NS['Ci'], 256, 256, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(nrot)]
    ys6 = [hip.empty_nhwc(NS['N'], NS['Co'], 256, 256, dev) for _ in range(nrot)]
    us6 = _time_conv(hip, xs6, w, b, ys6, iters)
    del xs6, ys6
    us_sus = _time_conv(hip, xs, w, b, ys, 4 * iters, warm_ms=20.0)       # last: the 240x240 launches again after 20 ms of back-to-back launches (the sustained rate)
    del xs, ys
    achieved = NS_BYTES / (us_rot * 1e-6) / 1e9
    traffic, traffic_src, kname = measure_traffic_inrun() if pmc_inrun else (None, None, None)
    pmc = os.path.join(ROOT, 'profiles', 'northstar_conv_pmc.json')
    if traffic is None and os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            traffic = rec.get('hbm_bytes_per_launch')
            traffic_src = (f"profiles/northstar_conv_pmc.json ({rec.get('tag', 'untagged')}, commit {rec.get('commit', 'n/a')}): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                           f"passes, STORED -- not measured in this run" + (f' (the in-run passes failed: {traffic_src})' if traffic_src else ''))
        except Exception:
            traffic = None
    gbs = lambda nbytes, us: round(nbytes / (us * 1e-6) / 1e9, 1)
    kernel = ('c4conv_split6_kernel<false, true> (six bf16 products per fp32 product, fp32-equivalent: DESIGN 4.1)' if six else 'c4conv_kernel<1, false, true> (fp32 MFMA)')
    return {'bound': 'hbm', 'kernel': kernel + ' -- 3x3 s1, 32x4x240x240 -> 32ch, fp32 NHWC', 'kernel_name_in_counter_file': kname,
            'fp32_mfma_form': None if us_f32 is None else {'kernel': 'c4conv_kernel<1, false, true> (option split6 = 0: v_mfma_f32_32x32x2_f32, exact fp32)', 'us_per_launch': round(us_f32, 2),
                                                           'achieved': gbs(NS_BYTES, us_f32), 'frac': round(gbs(NS_BYTES, us_f32) / HBM_PEAK_GBS, 4)},
            'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': traffic, 'traffic_source': traffic_src, 'algorithmic_bytes': NS_BYTES, 'us_per_launch': round(us_rot, 2),
            'timing': f'{iters} launches over {nrot} rotating x/y pairs (1.06 GB, beyond the 256 MiB Infinity Cache), HIP events (the protocol of every round: a ~1 ms burst)',
            'sustained': {'us_per_launch': round(us_sus, 2), 'achieved': gbs(NS_BYTES, us_sus), 'frac': round(gbs(NS_BYTES, us_sus) / HBM_PEAK_GBS, 4),
                          'note': f'{4 * iters} launches after 20 ms of back-to-back launches of the same kernel: under sustained store traffic the chip settles 10-25 % below the burst rate '
                                  '(rocprofv3 over 335 consecutive launches: 42.9 us min / 56.6 avg / 71.5 max, profiles/r06_northstar_sustained_kernel_stats.md); in the training step the kernel '
                                  'runs between other kernels: in_step_256'},
            'output_checked': f'vs torch fp32 conv2d on the host, max rel err {err:.1e}',
            'store_only_ceiling': {'gbs': round(fill_gbs, 1), 'us_for_this_output': round(us_fill, 2), 'kernel_time_over_it': round(us_rot / us_fill, 3),
                                   'note': 'mrdis_stream_fill: a kernel that ONLY writes the 236 MB output (non-temporal 16-byte stores, contiguous run per workgroup, nothing '
                                           'read, same rotating buffers, this run): the rate this device takes stores at.  The convolution writes the same bytes, reads 30 MB and '
                                           'computes in kernel_time_over_it x that time; the headline frac stays achieved / 8 TB/s'},
            'single_buffer': {'us_per_launch': round(us_one, 2), 'achieved': gbs(NS_BYTES, us_one), 'frac': round(gbs(NS_BYTES, us_one) / HBM_PEAK_GBS, 4)},
            'in_step_256': {'shape': '32x4x256x256 -> 32ch', 'algorithmic_bytes': bytes6, 'us_per_launch': round(us6, 2),
                            'achieved': gbs(bytes6, us6), 'frac': round(gbs(bytes6, us6) / HBM_PEAK_GBS, 4)},
            'tflops': round(2 * 9 * NS['Ci'] * NS['Co'] * NS['N'] * NS['H'] * NS['W'] / (us_rot * 1e-6) / 1e12, 2)}


def roofline_step(mrdis, dev, B, H, W, dtype, iters=6, only=None, pmc_inrun=False):
    """The three kernel families that dominate the step -- the fused gamma | beta convolutions of the full-, half- and
    quarter-resolution SPADE blocks (reference model.py:2443-2444; 16 calls each per step, forward + data gradient + weight
    gradient = ~40 % of the step) -- timed live with HIP events at the shapes of the timed step and priced against the MFMA peak
    of the arithmetic type.  fp32: Winograd F(2x2,3x3) executes 4/9 and F(4x4,3x3) 1/4 of the direct multiplies, so `frac` = direct-equivalent
    FLOPs x that factor / time / 157.3 TF (the kernels' own MFMA work against the matrix pipe); bf16: direct FLOPs / time / bf16 peak,
    and the HBM fraction of the layer's algorithmic bytes beside it (these layers are HBM-bound in bf16).
    `only` = (layer, entry): just that entry point (tools/step_kernel.py under rocprofv3 --pmc); `pmc_inrun`: the full-resolution F(4x4) entries (sp6
    fwd_spade, sp6 dgrad) also carry `traffic` = HBM bytes per launch from in-run PMC child passes, beside their algorithmic bytes."""
    hip = mrdis.hip
    bf = dtype != 'f32'
    out = []
    for name, ci, d in (('sp6.gamma+beta', 32, 1), ('sp5.gamma+beta', 64, 2), ('sp4.gamma+beta', 128, 4)):
        if only is not None and only[0] != name:
            continue
        co, h, w = 2 * ci, H // d, W // d
        el = torch.bfloat16 if dtype == 'bf16' else torch.float32
        x = torch.randn(B, ci, h, w, device=dev).to(el).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, co, h, w, device=dev).to(el).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(9, ci, co, device=dev) * 0.05
        wk = wt.permute(0, 2, 1).contiguous()
        bias = torch.zeros(co, device=dev)
        wb_f = hip.cast_bf16(wk) if bf else None
        wb_b = hip.cast_bf16(wt) if bf else None
        dt = hip.DT_F32_BF16M if bf else hip.DT_F32
        if not bf:      # as in the step: the filter's Winograd-domain images, built once per step behind the mixing launch (mrdis_wino_u_jobs)
            jobs, imgs, blocks = [], [], 0
            for src, R, S, flip in ((wt, ci, co, 0), (wk, co, ci, 1)):
                if S <= 32 and hip.wino_u_format(R, S) != 5:
                    imgs.append(None); continue
                img = torch.zeros(hip.wino_u_image_floats(R, S), device=dev)
                j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC = src.data_ptr(), img.data_ptr(), R, S, flip, 0
                j.block0, j.nblk = blocks, hip.wino_u_job_blocks(R, S); blocks += j.nblk
                jobs.append(j); imgs.append(img)
            if jobs:
                hip.wino_u_jobs(hip.wino_u_table(jobs, dev), len(jobs), blocks)
            wb_f, wb_b = imgs
        yo = hip.empty_nhwc(B, co, h, w, dev, el); dxo = hip.empty_nhwc(B, ci, h, w, dev, el)      # (outputs allocated once: the allocator is not timed)
        fns = {'fwd': lambda: hip.conv2d_fwd(x, wt, bias, 3, 3, 1, 1, w_bf16=wb_f, out=yo),
               'dgrad': lambda: hip.conv2d_bwd_data(dy, wk, (h, w), 3, 3, 1, 1, w_bf16=wb_b, out=dxo),
               'wgrad': lambda: hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, dtype=dt)}
        flop = 2.0 * 9 * ci * co * B * h * w
        if not bf and co == 2 * ci:
            # the forward AS THE STEP RUNS IT: the gamma | beta convolution with the SPADE modulation in its epilogue (model.py:2440-2446 as one kernel,
            # mrdis_conv2d_fwd_spade: z in, mix and gamma out, the 2C-channel tensor never written), statistics of z already taken by the resize in front
            C_ = ci
            simg = torch.zeros(hip.wino_u_image_floats(ci, co, C_), device=dev)
            j = hip.WinoUJob(); j.w, j.img, j.R, j.S, j.flip, j.spadeC = wt.data_ptr(), simg.data_ptr(), ci, co, 0, C_
            j.block0, j.nblk = 0, hip.wino_u_job_blocks(ci, co, C_)
            hip.wino_u_jobs(hip.wino_u_table([j], dev), 1, j.nblk)
            z_ = torch.randn(B, C_, h, w, device=dev).contiguous(memory_format=torch.channels_last)
            so = (hip.empty_nhwc(B, C_, h, w, dev, el), hip.empty_nhwc(B, C_, h, w, dev, el),
                  z_.mean(dim=(2, 3)).reshape(-1).contiguous(), (1.0 / (z_.var(dim=(2, 3), unbiased=False) + 1e-5).sqrt()).reshape(-1).contiguous())
            if hip.gb_spade_fwd(x, wt, bias, z_, 1e-5, out=so, stats_ready=True, w_wino=simg) is not None:
                fns['fwd_spade'] = lambda: hip.gb_spade_fwd(x, wt, bias, z_, 1e-5, out=so, stats_ready=True, w_wino=simg)

        def algo(R, S):        # which Winograd kernel the library's policy gives the forward / data gradient of an (R -> S) filter at this size
            nblk = B * ((h + 15) // 16) * ((w + 31) // 32) * ((S + 63) // 64)
            if not bf and hip.wino_u_format(R, S) == 4 and nblk >= 192:
                return 'winograd F(4x4,3x3)', 0.25
            if not bf and hip.wino_u_format(R, S) == 5 and B * ((h + 31) // 32) * ((w + 31) // 32) >= 192:
                regfed = hip.get_option('wino4r') != 0 and (R <= 64 or B * h * w * R * 4 > 300000000)
                if regfed or B * h * w * R * 4 <= 300000000:
                    return 'winograd F(4x4,3x3), 32-cout form' + (' (register-fed)' if regfed else ''), 0.25
            return 'winograd F(2x2,3x3)', 4.0 / 9.0
        w4 = (not bf) and ci % 32 == 0 and co % 64 == 0 and h % 8 == 0 and w % 8 == 0 and hip.get_option('wino4') != 0
        algos = {'fwd': algo(ci, co), 'dgrad': algo(co, ci),
                 'wgrad': ('winograd F(3x3,4x4)', 0.25) if w4 else ('winograd F(2x2,3x3)', 4.0 / 9.0)}
        if not bf and co == 2 * ci:
            s4 = hip.wino_u_format(ci, co, ci) == 4 and B * ((h + 15) // 16) * ((w + 31) // 32) * ((ci + 31) // 32) >= 192
            algos['fwd_spade'] = (('winograd F(4x4,3x3)' if s4 else 'winograd F(2x2,3x3)') + ' + SPADE modulation in the epilogue (the forward the step runs: z in, mix + gamma out)',
                                  0.25 if s4 else 4.0 / 9.0)
        nbytes = x.element_size() * x.numel() + dy.element_size() * dy.numel()
        row = {'layer': name, 'shape': f'{B}x{ci}x{h}x{w} -> {co}ch 3x3 s1', 'calls_per_step': 16, 'direct_gflop': round(flop / 1e9, 2),
               'algorithmic_bytes': nbytes}
        # what each entry point must move once (fp32): forward x + y; data gradient dy + dx; weight gradient x + dy; fused-SPADE forward x + z in, mix + gamma out
        esz = x.element_size()
        abytes = {'fwd': esz * B * h * w * (ci + co), 'dgrad': esz * B * h * w * (co + ci), 'wgrad': esz * B * h * w * (ci + co), 'fwd_spade': esz * B * h * w * 4 * ci}
        for k_, fn in fns.items():
            if only is not None and only[1] != k_:
                continue
            # >= 20 ms of warm-up per entry point (and >= 30 launches): from idle the first launches run ~20 % slower (497 vs 407 us on this
            # layer, tools/rs_probe.py) until the chip has ramped its clocks; inside the training step it has.  (Thirty launches of a 120 us
            # bf16 kernel were not enough: 161 us here against 118 us in tools/layer_bench.py.)
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record()
            for _ in range(30):
                fn()
            w1.record(); torch.cuda.synchronize()
            for _ in range(max(0, int(30 * (20.0 / max(w0.elapsed_time(w1), 0.1) - 1.0)))):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            if bf:
                row[k_] = {'us': round(us, 1), 'tflops': round(flop / us / 1e6, 1), 'frac_mfma': round(flop / us / 1e6 / MFMA_BF16_PEAK_TF, 3),
                           'gbs': round(nbytes / us / 1e3, 0), 'frac_hbm': round(nbytes / us / 1e3 / HBM_PEAK_GBS, 3)}
            else:
                name_, fac = algos[k_]
                row[k_] = {'us': round(us, 1), 'direct_equiv_tflops': round(flop / us / 1e6, 1), 'algorithm': name_,
                           'executed_multiplies_per_direct_multiply': round(fac, 4),
                           'frac_mfma': round(flop * fac / us / 1e6 / MFMA_F32_PEAK_TF, 3),
                           'algorithmic_bytes': abytes[k_], 'frac_hbm': round(abytes[k_] / us / 1e3 / HBM_PEAK_GBS, 3)}
                if pmc_inrun and only is None and name == 'sp6.gamma+beta' and k_ in ('fwd_spade', 'dgrad'):
                    tr, src, kn = measure_traffic_inrun(script=('step_kernel.py', name, k_, str(B), str(H), str(W)), kernel_substr='wino4', n_last=6,
                                                        what=f'tools/step_kernel.py {name} {k_}')
                    row[k_].update(traffic=tr, traffic_over_algorithmic=None if tr is None else round(tr / abytes[k_], 3), traffic_source=src, kernel_name_in_counter_file=kn)
        out.append(row)
        del x, dy, yo, dxo
    return {'bound': 'mfma' if not bf else 'hbm', 'peak': MFMA_F32_PEAK_TF if not bf else HBM_PEAK_GBS, 'unit': 'TFLOP/s' if not bf else 'GB/s',
            'pricing': 'fp32: the FLOPs the kernel EXECUTES on the matrix pipe = direct-equivalent FLOPs x 4/9 (Winograd F(2x2,3x3)) or x 1/4 (F(4x4,3x3), the 64+-channel '
                       'forward / data gradient) / time / 157.3 TF; direct_equiv_tflops is the same time priced at the direct convolution\'s FLOPs' if not bf else
                       'bf16: algorithmic bytes (x + dy in bf16) / time / 8 TB/s; direct FLOPs / time / 2500 TF beside it',
            'timing': f'{iters} launches per entry point after >= 20 ms / >= 30 launches of warm-up (clock ramp), HIP events on the launch stream, shapes of the timed step',
            'layers': out}


def cpu_baseline(M, H, W, adv):
    """The CPU restatement of the reference step (oracle, kind 'port'), bounded sample."""
    from oracle import ref_model as R
    import mrdis
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))        # the GPU box gives one GPU's job a 16-core share; more threads only thrash
    torch.set_num_threads(cores)
    print(f'[bench] cpu_baseline: {cores} threads, M={M}, {H}x{W}', file=sys.stderr, flush=True)
    B = 4
    torch.manual_seed(10); np.random.seed(10)
    model = R.RefMultimodalModel((H, W), M, is_discrim_s=adv).train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, weight_decay=1e-5, amsgrad=True)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, H, W, seed=10)
    lam = dict(R.DEFAULT_LAMBDAS, adv_s=1.0 if adv else 0.0)

    def one():
        loss, parts, _ = R.ref_forward_losses(model, x, mask, mask_img, lam)
        loss.backward(retain_graph=adv)
        gd = None
        if adv:
            g_main = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
            opt.zero_grad()
            parts['adv_s_d'].backward()
            for p, g in zip(model.parameters(), g_main):
                p.grad = g
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step(); opt.zero_grad()
    t0 = time.perf_counter()
    one()
    print(f'[bench] cpu_baseline warm-up step {time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        one()
        print(f'[bench] cpu_baseline step done at +{time.perf_counter() - t0:.1f} s', file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / n
    return {'value': round(B / dt, 4), 'unit': 'slices/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/ref_model.py train step, B={B}, M={M}, {H}x{W} fp32, 1 warm-up + {n} timed steps, '
                      f'{dt:.2f} s/step'}


def host_unblocked_ms(mrdis, cfg, dev, B, M, adv, steps=4, graph=False):
    """What the HOST needs to enqueue one step when the launch queue never fills: the same step (same launch count: it does not depend on the
    map size) on 64x64 slices, where the GPU finishes long before the host, timed per step between synchronisations.  `host_enqueue_ms_per_step` of the
    timed region is mostly back-pressure from the full queue; this number is what decides whether N ranks on one host stay GPU-bound."""
    small = dict(cfg); small.update(input_height=64, input_width=64)
    torch.manual_seed(10); np.random.seed(10)
    model = mrdis.build_model(small).train()
    step = mrdis.TrainStep(model, small)
    if graph:                                       # the same question for the graph-replayed step: closures re-drawn, one copy kernel, one graph launch
        step = mrdis.GraphedTrainStep(step, warm=1)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, 64, 64, seed=10)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last); maskd, mimgd = mask.to(dev), mask_img.to(dev)
    ts = []
    for i in range(steps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(xd, maskd, mimgd, mask)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    del model, step
    torch.cuda.empty_cache()
    return round(float(np.median(ts[2:])), 1)


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a CHILD `torch.distributed.run` (never exec: the parent has
    not touched a GPU and does not), forward its output, exit with its return code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(8, (os.cpu_count() or 1) // a.gpus))))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] --gpus {a.gpus} without a launcher: starting {a.gpus} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def dry_run(a, world, rank):
    """Launcher rehearsal (no GPU): the ranks rendezvous over gloo exactly as they would over RCCL and rank 0 prints who reported."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if 'MASTER_PORT' not in os.environ:            # only ever the single-process case (a launcher sets it): any free port, so concurrent runs on one host do not collide
        import socket
        assert world == 1, 'dry run with WORLD_SIZE > 1 needs the launcher\'s MASTER_PORT'
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = torch.tensor([rank], dtype=torch.int64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    if rank == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'rccl_ranks': dist.get_world_size(), 'backend': 'gloo (dry run)',
                          'ranks_reported': sorted(int(t) for t in got), 'steps': a.steps, 'warmup': a.warmup}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != a.gpus:
        print(f'bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE = {world} ranks; refusing to report a number for the wrong job '
              f'(under torchrun / torch.distributed.run pass --gpus {world} explicitly: the default is --gpus 1)', file=sys.stderr, flush=True)
        sys.exit(2)
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.dry_run:
        return dry_run(a, world, rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback for the hot path)'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if world > 1:
        # one host serves all ranks: give each rank its share of the cores (torch's default is every core per process,
        # which makes N ranks fight over the same cores while they enqueue ~3,000 launches per step)
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
        torch.set_num_threads(max(1, min(8, ncpu // max(1, local_world))))
    if world > 1:
        assert torch.cuda.device_count() >= world, f'--gpus {world} but only {torch.cuda.device_count()} GPUs are visible'
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == a.gpus
    import mrdis
    mrdis.hip.load()

    M, B = a.modalities, a.batch
    H, W = (256, 256) if a.fit == 'pad' else (160, 192)
    if a.slice != 240:
        H = W = (a.slice + 31) // 32 * 32
    adv = not a.no_adv
    cfg = dict(mrdis.DEFAULT_CONFIG)
    cfg.update(contrast_list=['T1', 'T1c', 'T2', 'T2_FLAIR'][:M] if M <= 4 else [f'm{i}' for i in range(M)],
               input_height=H, input_width=W, batch_size=B, lambda_adv_s=1.0 if adv else 0.0, compute_dtype=a.dtype)
    if a.recon_y:
        cfg.update(lambda_recon_y=1.0, out_num_ch=4)
    cfg = mrdis.derive_config(cfg, dev)
    # the step's dominant kernels at the step's shapes (before the model exists: 1.6 GB of operands of their own)
    # (world == 1 only: at N > 1 rank 0's extra legs run AFTER the process group is destroyed, so that no other rank waits in an RCCL barrier for them)
    rstep = roofline_step(mrdis, dev, B, H, W, a.dtype, pmc_inrun=(not a.no_pmc and a.dtype == 'f32')) if (world == 1 and not a.no_roofline) else None
    torch.cuda.empty_cache()
    if a.roofline_only:
        assert world == 1
        print(json.dumps({'roofline': roofline_conv(mrdis, dev), 'roofline_step': rstep}), flush=True)
        return
    torch.manual_seed(10); np.random.seed(10)                       # main_missing.py:18-21; same init on every rank
    model = mrdis.build_model(cfg).train()
    graph_on = a.graph or os.environ.get('MRDIS_GRAPH', '0') not in ('', '0')
    step = mrdis.TrainStep(model, cfg)
    eager_step = step
    if graph_on:
        step = mrdis.GraphedTrainStep(step)
        a.warmup = max(a.warmup, step.warm + 1)
    x, mask, mask_img = mrdis.synthetic_batch(B, M, a.slice, a.slice, seed=10 + rank, drop=a.drop)
    x = mrdis.fit_to_model(x, (H, W), fill=-10.0)
    mask_img = (x[:, 0] == 0).float()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    maskd, mimgd = mask.to(dev), mask_img.to(dev)
    tgt = torch.randint(0, 4, (B, 1, H, W), generator=torch.Generator().manual_seed(13 + rank)).float().to(dev) \
        if a.recon_y else None
    torch.manual_seed(100 + rank); np.random.seed(100)               # eps per rank, sim_s pair identical on all ranks

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)

    log(f'model built, batch {B}x{M}x{H}x{W}; warm-up')
    for i in range(a.warmup):
        step(xd, maskd, mimgd, mask, targets=tgt)
        torch.cuda.synchronize()
        log(f'warm-up step {i} done, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
    if step.reducer is not None:
        step.reducer.exposed_ms()                 # reset the exchange diagnostics (warm-up steps)
        step.reducer.timing = True
    sync()
    mrdis.hip.launch_counts(reset=True)
    t0 = time.perf_counter()
    host_ms = 0.0
    for _ in range(a.steps):
        h0 = time.perf_counter()
        loss, parts, _ = step(xd, maskd, mimgd, mask, targets=tgt)
        host_ms += (time.perf_counter() - h0) * 1e3
    torch.cuda.synchronize()
    lib_launches = mrdis.hip.launch_counts()['all'] / a.steps
    dt_local = time.perf_counter() - t0           # this rank's own time for the K steps (before the closing barrier)
    sync()
    dt = time.perf_counter() - t0
    host_ms /= a.steps
    ddp = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # what the exchange cost: per rank [own ms/step, host enqueue ms/step, compute-stream wait for the all-reduce ms/step]
        step.reducer.timing = False
        ex = step.reducer.exposed_ms()
        mine = torch.tensor([dt_local / a.steps * 1e3, host_ms, ex['exposed_ms'] / a.steps], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        ddp = {'per_rank_ms_per_step': [round(float(v), 2) for v in allr[:, 0]],
               'rank_ms_per_step_min': round(float(allr[:, 0].min()), 2), 'rank_ms_per_step_max': round(float(allr[:, 0].max()), 2),
               'host_enqueue_ms_per_step_per_rank': [round(float(v), 1) for v in allr[:, 1]],
               'allreduce_wait_ms_per_step_per_rank': [round(float(v), 3) for v in allr[:, 2]],
               'allreduce_wait_ms': round(float(allr[:, 2].max()), 3),
               'allreduce_wait_note': 'HIP events on the compute stream around the waits in GradAllReduce.finish(): time the compute stream '
                                      'idles for the gradient exchange per step (max over ranks); includes waiting for the slowest rank to arrive',
               'bytes_reduced_per_step': int(ex['bytes_reduced'] // max(1, a.steps)), 'buckets': ex['buckets'],
               'buckets_launched_during_backward_per_step': round(ex['early_buckets'] / max(1, a.steps), 2),
               'finish_calls_per_step': round(ex['finish_calls'] / max(1, a.steps), 2),
               'host_threads_per_rank': torch.get_num_threads()}
    ms = dt / a.steps * 1e3
    value = B * world / (dt / a.steps)
    host_losses = step.losses_to_host(parts)
    # the same step on the direct-convolution kernels only (north_star describes a direct conv; the default policy runs
    # fused Winograd on the big 3x3 layers): reported beside the headline, never as `value`
    ms_direct = None
    if not a.no_direct and a.dtype == 'f32' and mrdis.hip.get_option('wino') != 0:
        prev = mrdis.hip.get_option('wino')
        mrdis.hip.set_option('wino', 0)
        nd = max(2, min(a.steps, 5))
        eager_step(xd, maskd, mimgd, mask, targets=tgt)
        sync()
        t0 = time.perf_counter()
        for _ in range(nd):
            eager_step(xd, maskd, mimgd, mask, targets=tgt)
        sync()
        ms_direct = (time.perf_counter() - t0) / nd * 1e3
        mrdis.hip.set_option('wino', prev)
        if world > 1:
            t = torch.tensor([ms_direct], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_direct = float(t)

    # the four-channel thin layers with fp32 MFMA (option split6 = 0) instead of the six-product bf16 split the default policy uses for them: same run, same
    # box, a few steps -- beside the headline, with the loss both ways (the split is accepted only at the exact kernels' own error level: tests)
    split6 = None
    if a.dtype == 'f32' and not a.no_direct and mrdis.hip.get_option('split6') != 0:
        prev6 = mrdis.hip.get_option('split6')
        mrdis.hip.set_option('split6', 0)
        nd = max(2, min(a.steps, 5))
        eager_step(xd, maskd, mimgd, mask, targets=tgt)
        sync()
        t0 = time.perf_counter()
        for _ in range(nd):
            eager_step(xd, maskd, mimgd, mask, targets=tgt)
        sync()
        ms_off = (time.perf_counter() - t0) / nd * 1e3
        mrdis.hip.set_option('split6', prev6)
        if world > 1:
            t = torch.tensor([ms_off], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_off = float(t)
        split6 = {'ms_per_step_fp32_mfma_only': round(ms_off, 2),
                  'what': 'option split6 = 1 (default): the thin 3x3 stride-1 layers -- a four-channel side (4 -> C forward / C <- 4 data gradient up to 32 couts, C -> 4 forward / 4 <- C data gradient) and sp6.out (32 -> 16 forward, weight gradient) -- multiply on '
                          'v_mfma_f32_32x32x16_bf16 / 16x16x32 with both fp32 operands as three bf16 terms and the six products of order <= 2 summed in fp32 '
                          '(dropped: < 2^-23 of a product); = 0: fp32 MFMA.  Unit tests hold the split within 2e-6 of the fp32 kernels and at <= 2x their error against float64',
                  'steps_timed': nd}

    # the same step as HIP-graph replays (or, under --graph, eagerly): beside the headline, never as `value`
    graph_leg = None
    if not a.no_graph_leg and not a.no_direct and world == 1:
        try:
            nd = max(2, min(a.steps, 5))
            other = eager_step if graph_on else mrdis.GraphedTrainStep(eager_step, warm=1)
            for _ in range(3):                      # (graph: one eager iteration on the capture stream, the recording, one replay)
                other(xd, maskd, mimgd, mask, targets=tgt)
            sync()
            t0 = time.perf_counter(); h_ms = 0.0
            for _ in range(nd):
                h0 = time.perf_counter(); lg, _, _ = other(xd, maskd, mimgd, mask, targets=tgt); h_ms += (time.perf_counter() - h0) * 1e3
            sync()
            graph_leg = {'mode': 'eager' if graph_on else 'graph replay', 'ms_per_step': round((time.perf_counter() - t0) / nd * 1e3, 2),
                         'host_enqueue_ms_per_step': round(h_ms / nd, 2), 'steps_timed': nd, 'loss': round(float(lg), 5),
                         'stats': None if graph_on else dict(other.stats),
                         'what': 'trainer.GraphedTrainStep: forward, losses, backward(s), clip, Adam recorded into one HIP graph; per step the host re-draws eps / the '
                                 'sim_s + adv_s pairs / the mask weights (ops.host_value), ships them with one copy kernel and launches the graph.  Bit-identical to the eager '
                                 'step (tests/test_gpu_graph.py); opt-in (--graph, MRDIS_GRAPH=1, config graph: true)'}
        except Exception as ex:                     # noqa: BLE001 -- an extra leg must not take the headline down
            graph_leg = {'error': f'{type(ex).__name__}: {ex}'[:300]}
    if rank == 0:
        out = {
            'metric': 'MR slices/sec (train step, recon+adv+latent losses)' if adv else 'MR slices/sec (train step, recon+latent losses, lambda_adv_s=0)',
            'value': round(value, 3), 'unit': 'slices/s', 'n_gpus': world,
            'rccl_ranks': dist.get_world_size() if world > 1 else 1, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'bf16': 'bf16 (bf16 activations + MFMA operands, f32 accumulate / statistics / master weights)',
                      'bf16m': 'bf16 MFMA operands / f32 accumulate / f32 storage'}[a.dtype], 'data': 'synthetic',
            'config': {'workload': f'BraTS-shaped {M}-modality {a.slice}x{a.slice} fp32 slices '
                                   f'({"zero-padded with background to" if a.fit == "pad" else "centre-cropped to"} {H}x{W}), '
                                   f'batch {B}/GPU, full train step (fwd, recon_x+recon_x_mix+latent_z+sim_s+sim_z'
                                   f'{"+adv" if adv else ""}, bwd, clip, Adam amsgrad)',
                       'global_batch': B * world, 'per_gpu_batch': B, 'modalities': M, 'input_hw': [H, W],
                       'parallelism': f'dp{world}', 'missing_modality': bool(a.drop),
                       'output_decoder': bool(a.recon_y),
                       'conv_algorithms': {
                           'f32': 'fp32 throughout; direct MFMA kernels + fused Winograd for the big 3x3 stride-1 layers: F(4x4,3x3) forward / data gradient and '
                                  'F(3x3,4x4) weight gradient on the 64+-channel layers, F(2x2,3x3) on the rest '
                                  f'(options wino = {mrdis.hip.get_option("wino")}, wino4 = {mrdis.hip.get_option("wino4")}; wino = 0: direct only)',
                           'bf16': 'bf16 activations in HBM, direct convolutions on v_mfma_f32_32x32x16_bf16 (fp32 accumulate); fp32 kernels '
                                   'on the 4- / 7-channel boundary layers; fp32 statistics, losses, master weights, optimizer',
                           'bf16m': 'fp32 activations, bf16 MFMA operands (fp32 accumulate) in the direct convolution kernels'}[a.dtype]},
            'loss': round(host_losses['all'], 5),
            'step_tflops': round(FLOP_PER_SLICE_160x192 * (H * W) / (160 * 192) * (M / 4.0) ** 2 * B / (ms * 1e-3) / 1e12, 2),
            'step_tflops_note': 'direct-convolution-equivalent FLOPs / step time' + (
                ' (the big 3x3 layers run as Winograd F(4x4,3x3) / F(2x2,3x3): 1/4 / 4/9 of these multiplies are executed)' if a.dtype == 'f32' else ''),
            'mfma_peak_tflops': MFMA_F32_PEAK_TF if a.dtype == 'f32' else MFMA_BF16_PEAK_TF,
            'mfma_peak_dtype': 'f32' if a.dtype == 'f32' else 'bf16',
            'key_aliases': {'step_tflops_f32': 'step_tflops', 'mfma_f32_peak_tflops': 'mfma_peak_tflops', 'step_tflops_f32_direct_only': 'step_tflops_direct_only'},      # round-2 names of the same fields
            'ms_per_step_direct_only': None if ms_direct is None else round(ms_direct, 2),
            'split6': split6,
            'graph': {'headline_mode': 'graph replay' if graph_on else 'eager', 'other_mode': graph_leg},
            'step_tflops_direct_only': None if ms_direct is None else round(
                FLOP_PER_SLICE_160x192 * (H * W) / (160 * 192) * (M / 4.0) ** 2 * B / (ms_direct * 1e-3) / 1e12, 2),
        }
        log(f'timed: {ms:.1f} ms/step -> {value:.2f} slices/s (host enqueue {host_ms:.1f} ms/step)')
        out['library_launches_per_step'] = round(lib_launches, 1)
        out['library_launches_note'] = ('kernel launches issued by libmrdis_hip per timed step, counted by the library (mrdis_launch_count("all")); ATen glue kernels are not in it '
                                        '(profiles/*_dispatch_counts_*.txt has every dispatch from rocprofv3)')
        out['host_enqueue_ms_per_step'] = round(host_ms, 1)
        out['host_enqueue_note'] = 'host time inside step() during the timed region: mostly waiting on the full launch queue (the step is GPU-bound); host_ms_unblocked is the cost proper'
        if world == 1 and not a.no_roofline:
            out['host_ms_unblocked'] = host_unblocked_ms(mrdis, cfg, dev, B, M, adv)
            try:
                out['host_ms_unblocked_graph'] = host_unblocked_ms(mrdis, cfg, dev, B, M, adv, graph=True)
            except Exception as ex:                 # noqa: BLE001 -- an extra leg must not take the headline down
                out['host_ms_unblocked_graph'] = None
                out['host_ms_unblocked_graph_error'] = f'{type(ex).__name__}: {ex}'[:200]
            log(f'host_ms_unblocked: {out["host_ms_unblocked"]} ms/step')
        out['dynamic_lds_bytes'] = mrdis.hip.dynamic_lds()         # per kernel family: what rocprofv3's LDS column cannot show (tools/prof_summary.py merges it in)
        if ddp is not None:
            out['ddp'] = ddp
            out['allreduce_wait_ms'] = ddp['allreduce_wait_ms']
            out['bytes_reduced'] = ddp['bytes_reduced_per_step']
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()        # ranks 1..N-1 are done here; rank 0's extra legs below keep nobody waiting
    if rank == 0:
        if not a.no_roofline:
            if rstep is None:
                rstep = roofline_step(mrdis, dev, B, H, W, a.dtype)
            out['roofline'] = roofline_conv(mrdis, dev, pmc_inrun=(world == 1 and not a.no_pmc))
            log(f'roofline: {out["roofline"]}')
            out['roofline_step'] = rstep
            log(f'roofline_step: {rstep}')
            if world > 1:
                out['roofline_note'] = 'N > 1: both roofline legs ran on rank 0 after destroy_process_group() (outside the timed region, no rank waiting); PMC traffic passes are N = 1 only'
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(M, H, W, adv)
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
