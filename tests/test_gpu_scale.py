"""Parity AT THE BENCHMARKED SCALE (BASELINE.json configs[1]: B=32, M=4, 240x240 padded to 256x256, fp32).

The small-shape tests of test_gpu_ops.py cannot reach the code paths that only exist at this size: the grid-size
policy that picks the Winograd kernels (option "wino" = 1, the default), their non-temporal-store epilogue (outputs
>= 128 MB, csrc/mrdis_wino.hip `nt_out`), the 256-position tiles of the direct kernel, the persistent Cin = 4 kernel
beyond the Infinity Cache.  Every conv geometry of the step is run here at the batch the step uses (SPADENewShared
runs at 4B = 128) under the default policy and checked
  * against torch fp32 on the CPU on a sampled sub-batch (forward, data gradient; the weight gradient through a
    cotangent that is zero outside the sampled images, which every workgroup still walks), and
  * against the direct kernels (wino = 0) on the full tensors (all three directions).
Reference geometry: /root/reference/src/model.py:2218-2296 (U-Net), :2332-2400 (modality encoder), :2424-2454,
:2584-2632 (SPADE blocks), :2769-2800 (discriminator).  Tolerance: 1e-3 relative (north_star); asserted tighter."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fixtures import dump_measured   # noqa: E402

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
B, HW = 32, 256


def zoo():
    """(name, N, Ci, Co, k, stride, pad, H, W) -- every distinct convolution of one B=32, M=4, 256x256 step as the
    step issues it (gamma+beta fused into one Co = 2 Ci convolution; sp1-sp3 on the 4B batch)."""
    c, H = 32, HW
    L = [('ana.down_1', B, 7, c, 4, 2, 1, H, H), ('ana.down_2', B, c, 2 * c, 4, 2, 1, H // 2, H // 2),
         ('ana.down_3', B, 2 * c, 4 * c, 4, 2, 1, H // 4, H // 4), ('ana.down_4', B, 4 * c, 8 * c, 4, 2, 1, H // 8, H // 8),
         ('ana.down_5', B, 8 * c, 8 * c, 4, 2, 1, H // 16, H // 16),
         ('ana.up_4', B, 8 * c, 8 * c, 3, 1, 1, H // 16, H // 16), ('ana.up_3', B, 16 * c, 4 * c, 3, 1, 1, H // 8, H // 8),
         ('ana.up_2', B, 8 * c, 2 * c, 3, 1, 1, H // 4, H // 4), ('ana.up_1', B, 4 * c, c, 3, 1, 1, H // 2, H // 2),
         ('ana.output', B, 2 * c, 4, 3, 1, 1, H, H)]
    h = H
    for i, (a, b) in enumerate([(7, 16), (16, 32), (32, 64), (64, 128), (128, 128)]):
        L.append((f'mod.conv{i + 1}', B, a, b, 3, 2, 1, h, h)); h //= 2
    for i, (ci, co, d, n) in enumerate([(128, 128, 32, 4 * B), (128, 128, 16, 4 * B), (128, 128, 8, 4 * B),
                                        (128, 64, 4, B), (64, 32, 2, B), (32, 16, 1, B)]):
        hh = H // d
        L.append((f'sp{i + 1}.si', n, 4, ci, 3, 1, 1, hh, hh))
        L.append((f'sp{i + 1}.gamma+beta', n, ci, 2 * ci, 3, 1, 1, hh, hh))
        L.append((f'sp{i + 1}.out', n, ci, co, 3, 1, 1, hh, hh))
    L.append(('dec.out1x1', B, 16, 7, 1, 1, 0, H, H))
    h = H
    for i, (a, b) in enumerate([(4, 16), (16, 32), (32, 64), (64, 128), (128, 64)]):
        L.append((f'disc.conv{i + 1}', B, a, b, 4, 2, 1, h, h)); h //= 2
    return L


ZOO = zoo()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def cl(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last)


def rel(got, want):
    got = got.detach().float().cpu(); want = want.detach().float().cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    return float((got - want).abs().max()) / (float(want.abs().max()) + 1e-30)


def to_tck(w):
    Co, Ci, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw, Ci, Co).contiguous()


def to_tkc(w):
    Co, Ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, Co, Ci).contiguous()


@pytest.mark.parametrize('case', ZOO, ids=[c[0] for c in ZOO])
def test_layer_zoo_at_bench_scale(mrdis, case):
    name, N, Ci, Co, k, s, p, H, W = case
    hip = mrdis.hip
    assert hip.get_option('wino') == 1                    # the policy the benchmark runs under
    S = sorted({0, N // 2 + 1, N - 1})                    # sampled images for the CPU reference
    w = rnd((Co, Ci, k, k), 2, 0.5 / np.sqrt(Ci * k * k)); b = rnd((Co,), 3, 0.1)
    Ho, Wo = hip.conv_out_hw(H, W, k, k, s, p)
    gen = torch.Generator(device=DEV).manual_seed(1)      # the big tensors are drawn on the device; the samples go to the host
    xd = torch.randn((N, Ci, H, W), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    dyd = torch.randn((N, Co, Ho, Wo), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    bd = b.to(DEV)
    w_tck, w_tkc = to_tck(w).to(DEV), to_tkc(w).to(DEV)
    # CPU fp32 on the sampled images
    xs = xd[S].cpu().contiguous().requires_grad_(True); ws = w.clone().requires_grad_(True)
    dy_s = dyd[S].cpu().contiguous()
    ys = F.conv2d(xs, ws, b, s, p)
    ys.backward(dy_s)
    # default policy
    y = hip.conv2d_fwd(xd, w_tck, bd, k, k, s, p)
    dx = hip.conv2d_bwd_data(dyd, w_tkc, (H, W), k, k, s, p)
    dw, db = hip.conv2d_bwd_weight(xd, dyd, k, k, s, p, need_bias=True)
    assert rel(y[S], ys) <= 1e-4, ('fwd vs torch', name)
    assert rel(dx[S], xs.grad) <= 1e-4, ('dgrad vs torch', name)
    dys = torch.zeros_like(dyd); dys[S] = dyd[S]
    dw_s, db_s = hip.conv2d_bwd_weight(xd, dys, k, k, s, p, need_bias=True)
    assert rel(dw_s, to_tck(ws.grad)) <= 2e-4, ('wgrad vs torch', name)
    assert rel(db_s, dy_s.sum((0, 2, 3))) <= 2e-4, ('dbias vs torch', name)
    del dys
    # direct kernels on the full tensors
    hip.set_option('wino', 0)
    y0 = hip.conv2d_fwd(xd, w_tck, bd, k, k, s, p)
    dx0 = hip.conv2d_bwd_data(dyd, w_tkc, (H, W), k, k, s, p)
    dw0, db0 = hip.conv2d_bwd_weight(xd, dyd, k, k, s, p, need_bias=True)
    hip.set_option('wino', 1)
    for what, a, c, tol in (('fwd', y, y0, 1e-4), ('dgrad', dx, dx0, 1e-4), ('wgrad', dw, dw0, 2e-4), ('dbias', db, db0, 2e-4)):
        err = float((a - c).abs().max()) / float(c.abs().max())
        assert err <= tol, (what + ' default policy vs direct', name, err)
    # the narrow layers (<= 16 channels on one side, stride-2 first layers, 1x1 head) have dedicated kernels (mrdis_c16 / co4 / pointwise /
    # wgrad_s2 / wgrad16): the generic tile kernels (option debug_now16 = 1) on the full tensors
    if min(Ci, Co) <= 16:
        hip.set_option('debug_now16', 1)
        try:
            y1 = hip.conv2d_fwd(xd, w_tck, bd, k, k, s, p)
            dx1 = hip.conv2d_bwd_data(dyd, w_tkc, (H, W), k, k, s, p)
            dw1, db1 = hip.conv2d_bwd_weight(xd, dyd, k, k, s, p, need_bias=True)
        finally:
            hip.set_option('debug_now16', 0)
        for what, a, c, tol in (('fwd', y, y1, 1e-4), ('dgrad', dx, dx1, 1e-4), ('wgrad', dw, dw1, 2e-4), ('dbias', db, db1, 2e-4)):
            err = float((a - c).abs().max()) / float(c.abs().max())
            assert err <= tol, (what + ' dedicated narrow kernel vs tile kernel', name, err)
        del y1, dx1, dw1, db1
    # the fused LeakyReLU epilogue at this size
    yl = hip.conv2d_fwd(xd, w_tck, bd, k, k, s, p, lrelu=True)
    assert torch.equal(yl, F.leaky_relu(y, 0.2)) or rel(yl, F.leaky_relu(y, 0.2)) <= 1e-6, ('lrelu', name)


def test_winograd_nontemporal_store_path_runs(mrdis):
    """outputs >= 128 MB leave wino_conv_kernel through the non-temporal-store epilogue (csrc/mrdis_wino.hip nt_out):
    same kernel with the threshold moved (option "nt_mb") must give bit-identical results."""
    hip = mrdis.hip
    N, Ci, Co, H = B, 32, 64, HW                          # sp6.gamma+beta: 537 MB output
    x = cl(rnd((N, Ci, H, H), 7)); w = to_tck(rnd((Co, Ci, 3, 3), 8, 0.05)).to(DEV); b = rnd((Co,), 9).to(DEV)
    y_nt = hip.conv2d_fwd(x, w, b, 3, 3, 1, 1)
    hip.set_option('nt_mb', 1 << 40)
    y_plain = hip.conv2d_fwd(x, w, b, 3, 3, 1, 1)
    hip.set_option('nt_mb', 128)
    hip.set_option('wino', 0)
    y_direct = hip.conv2d_fwd(x, w, b, 3, 3, 1, 1)
    assert torch.equal(y_nt, y_plain)
    assert not torch.equal(y_nt, y_direct)                # the default policy really ran the Winograd kernel here
    assert float((y_nt - y_direct).abs().max()) <= 1e-4 * float(y_direct.abs().max())


@pytest.mark.parametrize('N,C,h', [(32, 32, 128), (32, 64, 64), (128, 32, 128)])
def test_spade_backward_with_resize_adjoint_at_bench_scale(mrdis, N, C, h):
    """The SPADE backward of the benchmarked step's full-resolution blocks (one label: N = 32; the batch-concatenated decoders: N = 128 -- 8.6 G elements of traffic):
    the one-pass form (mrdis_instnorm_spade_bwd_up2 with xlo: U^T dzh + per-tile partial sums, finished on the low-resolution map) against the kernels it
    replaces (statistics pass + apply pass + the resize's adjoint): [d gamma | d beta] within 2e-6, d x within 2e-5 of the maximum; the step-level comparisons
    (default policy vs direct kernels) run the fused form on both sides, so this is the check that is independent of it."""
    hip = mrdis.hip
    H = 2 * h
    x = cl(rnd((N, C, h, h), 41))
    z = hip.bilinear_fwd(x, (H, H), False)
    gamma = cl(rnd((N, C, H, H), 42, 0.3))
    dout = cl(rnd((N, C, H, H), 43))
    mean = torch.stack([z[i:i + 8].mean(dim=(2, 3)) for i in range(0, N, 8)]).reshape(-1).contiguous()
    var = torch.stack([z[i:i + 8].var(dim=(2, 3), unbiased=False) for i in range(0, N, 8)]).reshape(-1)
    rstd = (1.0 / (var + 1e-5).sqrt()).contiguous()
    dz, dgb_ref = hip.instnorm_spade_bwd(dout, z, gamma, mean, rstd, fused_gb=True)
    dx_ref = hip.bilinear_bwd(dz, (h, h), False)
    del dz
    dx, dgb = hip.instnorm_spade_bwd(dout, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x)
    e_g = float((dgb - dgb_ref).abs().max() / dgb_ref.abs().max())
    e_x = float((dx - dx_ref).abs().max() / dx_ref.abs().max())
    assert e_g <= 2e-6 and e_x <= 2e-5, (e_g, e_x)


def test_north_star_conv_output(mrdis):
    """BASELINE.json north star: 3x3 s1 conv, x (32,4,240,240) -> (32,32,240,240) fp32, every output value vs torch."""
    hip = mrdis.hip
    x = rnd((32, 4, 240, 240), 11); w = rnd((32, 4, 3, 3), 12, 0.2); b = rnd((32,), 13, 0.1)
    want = F.conv2d(x, w, b, 1, 1)
    got = hip.conv2d_fwd(cl(x), to_tck(w).to(DEV), b.to(DEV), 3, 3, 1, 1)
    assert rel(got, want) <= 1e-5
    x6 = rnd((32, 4, 256, 256), 14)                        # the same layer at the in-step size (268 MB out)
    want6 = F.conv2d(x6, w, b, 1, 1)
    got6 = hip.conv2d_fwd(cl(x6), to_tck(w).to(DEV), b.to(DEV), 3, 3, 1, 1)
    assert rel(got6, want6) <= 1e-5


def test_full_step_at_bench_scale_winograd_vs_direct(mrdis):
    """One B=32, M=4, 256x256 training step (the headline workload, adversarial loss on) under the default policy vs
    (a) the direct kernels only (wino = 0) and (b) the generic tile kernels on the narrow layers (debug_now16 = 1: the dedicated
    c16 / co4 / pointwise / stride-2 / wgrad16 kernels off): loss, every loss part, every parameter gradient, and the weights after Adam."""
    m = mrdis
    H = W = HW
    res = {}
    # (wino, debug_now16[, tag]): default policy (F(4x4,3x3) on the wide 3x3 layers, F(2x2,3x3) elsewhere) | direct kernels only | narrow-layer kernels
    # off | F(2x2) only (wino4 = 0) with the filter images | F(2x2) only, filters transformed in the kernels (wino_u = 0)
    for mode in ((1, 0), (0, 0), (1, 1), (1, 0, 'f2'), (1, 0, 'f2_no_u')):
        m.hip.set_option('wino', mode[0]); m.hip.set_option('debug_now16', mode[1])
        m.hip.set_option('wino4', 0 if len(mode) > 2 else 1); m.hip.set_option('wino_u', 0 if mode[2:] == ('f2_no_u',) else 1)
        cfg = dict(m.DEFAULT_CONFIG); cfg.update(input_height=H, input_width=W, batch_size=B, lambda_adv_s=1.0, is_patch_gan=True)
        cfg = m.derive_config(cfg, DEV)
        torch.manual_seed(10); np.random.seed(10)
        model = m.build_model(cfg).train()
        x, mask, mask_img = m.synthetic_batch(B, 4, 240, 240, seed=10)
        x = m.fit_to_model(x, (H, W), fill=-10.0); mask_img = (x[:, 0] == 0).float()
        step = m.TrainStep(model, cfg)
        torch.manual_seed(11); np.random.seed(11)
        grads = {}
        # first step: the arena does not exist yet, so every gradient passes through autograd's accumulation; with the
        # adversarial loss there are two backward passes (generator, then discriminator loss) -> keep both
        hooks = [p.register_post_accumulate_grad_hook(lambda p_, n=n: grads.setdefault(n, []).append(p_.grad.detach().clone()))
                 for n, p in model.named_parameters()]
        loss, parts, _ = step(cl(x), mask.to(DEV), mask_img.to(DEV), mask)
        for h in hooks:
            h.remove()
        res[mode] = (float(loss), {k_: float(v) for k_, v in parts.items()}, grads,
                     {n: p.detach().clone() for n, p in model.named_parameters()})
        del model, step
        torch.cuda.empty_cache()
    m.hip.set_option('wino', 1); m.hip.set_option('debug_now16', 0); m.hip.set_option('wino_u', 1); m.hip.set_option('wino4', 1)
    # the pipelined F(2x2) kernels read the filter's 16-point image built behind the mixing launch, or transform the nine taps themselves
    # (option wino_u = 0): the same expressions on the same values -- the whole step is bit-identical
    (l1, p1, g1, w1), (l0, p0, g0, w0) = res[(1, 0, 'f2')], res[(1, 0, 'f2_no_u')]
    assert l1 == l0 and p1 == p0
    assert all(torch.equal(a, c) for n in g0 for a, c in zip(g0[n], g1[n])) and all(torch.equal(w0[n], w1[n]) for n in w0)
    assert any(not torch.equal(res[(1, 0)][2][n][0], g1[n][0]) for n in g1), 'option wino4 did not change any kernel'
    for other in ((0, 0), (1, 1), (1, 0, 'f2')):
        (l1, p1, g1, w1), (l0, p0, g0, w0) = res[(1, 0)], res[other]
        assert np.isfinite(l1) and abs(l1 - l0) <= 1e-4 * abs(l0), (l1, l0)
        for k_ in p0:
            assert abs(p1[k_] - p0[k_]) <= 2e-4 * abs(p0[k_]) + 1e-7, (k_, p1[k_], p0[k_])
        assert set(g0) == set(g1) and len(g0) > 200
        assert any(not torch.equal(g0[n][0], g1[n][0]) for n in g0), 'the default policy did not change any kernel'
        for pass_ in (0, 1):
            names = [n for n in g0 if len(g0[n]) > pass_]
            tot = float(torch.sqrt(sum((g0[n][pass_].double() ** 2).sum() for n in names)))
            for n in names:
                a, c = g0[n][pass_], g1[n][pass_]
                err = float((a - c).double().norm())
                assert err <= 2e-3 * float(a.double().norm()) + 2e-5 * tot, (pass_, n, err, float(a.norm()))
        # an Adam step moves a weight by at most ~lr = 2e-4 (the sign of a noise-level gradient, e.g. of a conv bias in front of
        # BatchNorm, may flip): bound the worst case
        # and require the bulk to agree
        for n in w0:
            d = (w0[n] - w1[n]).abs()
            assert float(d.max()) <= 8.4e-4, n          # two Adam steps (generator + discriminator optimizer), each <= ~lr per element
        num = sum(float((w0[n] - w1[n]).abs().sum()) for n in w0); den = sum(w0[n].numel() for n in w0)
        assert num / den <= 2e-6, num / den


def _bf16_path(Ci, Co, s, k=3):
    """which kernels ops.conv2d runs a layer on under compute_dtype 'bf16': 'bf16' = the bf16 MFMA kernels on bf16 views (a 4- / 7-channel
    side of a stride-1 layer zero-padded to 16), 'f32' = the fp32 kernels between view casts (the stride-2 first layers), 'head' = the
    1x1 decoder head on the mixed-storage streaming kernels (bf16 in, fp32 weights and out)"""
    if k == 1 and Ci == 16 and Co <= 8:
        return 'head'
    if Ci % 16 == 0 and Co % 4 == 0 and Co >= 16:
        return 'bf16'
    return 'bf16' if s == 1 else 'f32'


def _run_bf16_layer(m, xd, w, b, dyd, k, s, p, Co, prepadded=False):
    """the layer as the bf16 step runs it: ops.conv2d (policy + view casts + autograd) inside a mixed-kernel cache scope.
    prepadded: the filter arrives zero-padded to 16 channels, as the all-layers mixing launch writes the narrow ones (MixPlan.padded)."""
    ops = m.ops
    Ci = w.shape[1]
    w_tck = to_tck(w)
    if prepadded:
        w_tck = F.pad(w_tck, (0, max(Co, 16) - Co, 0, max(Ci, 16) - Ci))
    w_tck = w_tck.to(DEV).requires_grad_(True)
    w_tkc = w_tck.detach().permute(0, 2, 1).contiguous()
    bias = b.to(DEV).requires_grad_(True)
    x = xd.detach().requires_grad_(True)
    with ops.mix_cache():
        y = ops.conv2d(x, w_tck, w_tkc, bias, k, k, s, p, co=Co)
        y.backward(dyd.to(y.dtype))
    dw = w_tck.grad[:, :Ci, :Co] if prepadded else w_tck.grad
    return y.detach(), x.grad, dw, bias.grad[:Co]


@pytest.mark.parametrize('case', ZOO, ids=[c[0] for c in ZOO])
def test_layer_zoo_at_bench_scale_bf16(mrdis, case):
    """BASELINE configs[2] at the benchmarked scale: every convolution of the step at B = 32 / 256x256 as `compute_dtype: bf16` runs it
    (ops.conv2d: bf16 MFMA kernels on bf16 views -- persistent bconv3 / bwgrad2 workgroups walking hundreds of items, the packed
    stride-2 data gradient, narrow sides zero-padded to 16 -- or the fp32 kernels between view casts on the stride-2 first layers):
      * forward / data gradient on sampled images and the weight / bias gradient (cotangent zero outside them) against torch fp32
        on the operands the kernels actually multiply (bf16-rounded where the bf16 kernels run), up to the rounding of a bf16 result;
      * option wino_pipe 0 vs 1 (bconv3 / bwgrad2 vs bconv / bwgrad) and debug_nopack 0 vs 1 (packed parity classes): bit-identical
        on the full tensors;
      * a filter that arrives pre-padded to 16 channels (the mixing launch's layout) against the explicit pad of ops.conv2d: bit-identical.
    Reference geometry as in test_layer_zoo_at_bench_scale."""
    name, N, Ci, Co, k, s, p, H, W = case
    m, hip = mrdis, mrdis.hip
    B16 = torch.bfloat16
    path = _bf16_path(Ci, Co, s, k)
    S = sorted({0, N // 2 + 1, N - 1})
    w = rnd((Co, Ci, k, k), 2, 0.5 / np.sqrt(Ci * k * k)); b = rnd((Co,), 3, 0.1)
    Ho, Wo = hip.conv_out_hw(H, W, k, k, s, p)
    gen = torch.Generator(device=DEV).manual_seed(1)
    xd = torch.randn((N, Ci, H, W), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    dyd = torch.randn((N, Co, Ho, Wo), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
    if Ci >= 16:
        xd = xd.to(B16)                                   # stored in bf16 by its producer
    if Co >= 16:
        dyd = dyd.to(B16)
    m.ops.set_compute_dtype('bf16')
    try:
        y, dx, dw, db = _run_bf16_layer(m, xd, w, b, dyd, k, s, p, Co)
        assert y.dtype == (B16 if Co >= 16 else torch.float32) and dx.dtype == xd.dtype and dw.dtype == torch.float32
        # torch fp32 on what the kernels multiply
        r16 = lambda t_: t_.float().bfloat16().float()
        xs = xd[S].float().cpu(); dy_s = dyd[S].float().cpu(); wr = w
        if path == 'bf16':
            xs, dy_s, wr = r16(xs), r16(dy_s), r16(w)
        xs = xs.contiguous().requires_grad_(True); ws = wr.clone().requires_grad_(True)
        ys = F.conv2d(xs, ws, b, s, p)
        ys.backward(dy_s)
        if Ci == 4 and path == 'bf16':
            # the si_layers' forward runs on the Cin = 4 kernel (fp32 map and filter in, bf16 out); only their backward multiplies bf16-rounded operands
            assert rel(y[S], F.conv2d(xd[S].float().cpu(), w, b, s, p)) <= 4e-3, ('fwd (Cin = 4 kernel) vs torch', name)
        else:
            assert rel(y[S], ys) <= 4e-3, ('fwd vs torch', name, rel(y[S], ys))                   # a bf16 result: 2^-9 relative per element
        dx_ref = xs.grad
        if Co == 4 and k == 3 and s == 1 and path == 'bf16':
            # the C -> 4 layer's data gradient runs on the Cin = 4 kernel: fp32 dy and fp32 filter in, bf16 out (MRDIS_DT_XBF16_YF32)
            dx_ref = torch.nn.grad.conv2d_input(tuple(xs.shape), w, dyd[S].float().cpu().contiguous(), s, p)
        assert rel(dx[S], dx_ref) <= 4e-3, ('dgrad vs torch', name, rel(dx[S], dx_ref))
        dys = torch.zeros_like(dyd); dys[S] = dyd[S]
        _, _, dw_s, db_s = _run_bf16_layer(m, xd, w, b, dys, k, s, p, Co)
        wg_ref = ws.grad
        if Ci == 4 and path == 'bf16':
            # ... and so does their weight gradient (wgrad_c4_kernel: fp32 map x bf16 output gradient; maps narrower than 64: the fp32 kernel on dy.float())
            w4 = w.clone().requires_grad_(True)
            F.conv2d(xd[S].float().cpu().contiguous(), w4, b, s, p).backward(dy_s)
            wg_ref = w4.grad
        if Co == 4 and k == 3 and s == 1 and path == 'bf16' and W in (64, 128, 256):
            # the C -> 4 layer's weight gradient multiplies the bf16 trunk with the fp32 gradient itself (wgrad_c4b_kernel<.., SWAP>, MRDIS_DT_XBF16_YF32)
            w4 = w.clone().requires_grad_(True)
            dy_s = dyd[S].float().cpu().contiguous()
            F.conv2d(xd[S].float().cpu().contiguous(), w4, b, s, p).backward(dy_s)
            wg_ref = w4.grad
        assert rel(dw_s, to_tck(wg_ref)) <= 4e-4, ('wgrad vs torch', name, rel(dw_s, to_tck(wg_ref)))   # fp32 accumulation of bf16 products
        assert rel(db_s, dy_s.sum((0, 2, 3))) <= 4e-4, ('dbias vs torch', name)
        del dys, dw_s, db_s
        # pipelined vs plain bf16 kernels, packed vs four-launch stride-2 data gradient: same arithmetic in the same order
        for opt, val in (('wino_pipe', 0), ('debug_nopack', 1)):
            with hip.option(opt, val):
                y1, dx1, dw1, db1 = _run_bf16_layer(m, xd, w, b, dyd, k, s, p, Co)
            for what, a, c in (('fwd', y, y1), ('dgrad', dx, dx1), ('wgrad', dw, dw1), ('dbias', db, db1)):
                assert torch.equal(a, c), (f'{what}: option {opt} = {val} changed the result', name, rel(a, c))
            del y1, dx1, dw1, db1
        if path == 'bf16' and min(Ci, Co) < 16:
            y2, dx2, dw2, db2 = _run_bf16_layer(m, xd, w, b, dyd, k, s, p, Co, prepadded=True)
            for what, a, c in (('fwd', y, y2), ('dgrad', dx, dx2), ('wgrad', dw, dw2), ('dbias', db, db2)):
                assert torch.equal(a, c), (f'{what}: pre-padded filter vs explicit pad', name, rel(a, c))
    finally:
        m.ops.set_compute_dtype('f32')


def _one_step(m, mode, B_=B, H=HW, W=HW, opts=(), drop=None, extra=None):
    """loss, loss parts, every parameter gradient (both backward passes) and the weights after Adam of ONE headline step.
    drop: None | 'random' (BASELINE configs[3]: one random modality missing per slice, util.py:538-542) | 'absent3' (modality 3 missing from the
    WHOLE batch, another one at random from every second slice); extra: dict that receives the weights before the step and the step's active decoders"""
    for o, v in opts:
        m.hip.set_option(o, v)
    try:
        cfg = dict(m.DEFAULT_CONFIG); cfg.update(input_height=H, input_width=W, batch_size=B_, lambda_adv_s=1.0, is_patch_gan=True, compute_dtype=mode)
        cfg = m.derive_config(cfg, DEV)
        torch.manual_seed(10); np.random.seed(10)
        model = m.build_model(cfg).train()
        x, mask, mask_img = m.synthetic_batch(B_, 4, 240, 240, seed=10, drop=(drop == 'random'))
        if drop == 'absent3':
            g = torch.Generator().manual_seed(77)
            mask[:, 3] = 0; x[:, 21:28] = 0
            for b_ in range(0, B_, 2):
                d_ = int(torch.randint(0, 3, (1,), generator=g))
                mask[b_, d_] = 0; x[b_, 7 * d_:7 * d_ + 7] = 0
        x = m.fit_to_model(x, (H, W), fill=-10.0); mask_img = (x[:, 0] == 0).float()
        step = m.TrainStep(model, cfg)
        if extra is not None:
            extra['w_before'] = step.optimizer.flat_p.detach().clone()
            extra['active'] = model.active_decoders(mask.numpy())
        torch.manual_seed(11); np.random.seed(11)
        loss, parts, _ = step(cl(x), mask.to(DEV), mask_img.to(DEV), mask)
        if extra is not None:
            extra['m_after'] = step.optimizer.m.detach().clone(); extra['skipped'] = step.optimizer.skipped_steps()
        names = {id(p): n for n, p in model.named_parameters()}
        # the arena exists from the constructor: after the step the gradient buffers are zeroed, so read what the clip saw instead
        gnorm = float(step.last_grad_norm_sq[0].sqrt())
        out = (float(loss), {k_: float(v) for k_, v in parts.items()}, gnorm, step.optimizer.flat_p.detach().clone(),
               [names[id(p)] for p in step.optimizer.used], list(step.optimizer.offsets))
        del model, step
        torch.cuda.empty_cache()
        return out
    finally:
        for o, v in opts:
            m.hip.set_option(o, {'wino_pipe': 1, 'debug_nopack': 0, 'wino': 1, 'wino4': 1}[o])
        m.ops.set_compute_dtype('f32')


def test_bf16_boundary_kernels_at_bench_scale(mrdis):
    """The two ends of the decoders' bf16 stretch as the grouped decoders run them at B = 32 / 256x256 (ops.conv2d_grouped, two groups):
    the 4 -> 32 si_layers with the filter in the 16-row layout of the mixing launch -- forward on the Cin = 4 kernel reading the fp32
    anatomy map and writing bf16 (MRDIS_DT_XF32_YBF16), backward on the padded bf16 kernels -- and the 1x1 head 16 -> 7 on the
    mixed-storage streaming kernels (MRDIS_DT_XBF16_YF32: bf16 trunk in, fp32 reconstruction out, all three directions).  Against torch
    fp32 on the operands each kernel multiplies, and the si forward against the padded bf16 MFMA path (option debug_noc4)."""
    m, hip, ops = mrdis, mrdis.hip, mrdis.ops
    B16 = torch.bfloat16
    N, H = B, HW
    S = [0, N - 1]
    r16 = lambda t_: t_.float().bfloat16().float()
    gen = torch.Generator(device=DEV).manual_seed(5)
    ops.set_compute_dtype('bf16')
    try:
        # ---- si_layers: one shared fp32 input, two filter sets
        x = torch.randn((N, 4, H, H), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
        ws = [rnd((32, 4, 3, 3), 20 + g, 0.2) for g in range(2)]; b = rnd((32,), 23, 0.1)
        dy = torch.randn((2 * N, 32, H, H), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last).to(B16)

        def run_si():
            xs = x.detach().requires_grad_(True)
            tcks = [F.pad(to_tck(w), (0, 0, 0, 12)).to(DEV).requires_grad_(True) for w in ws]          # [9][16][32], rows 4.. zero
            bias = b.to(DEV).requires_grad_(True)
            with ops.mix_cache():
                y = ops.conv2d_grouped(xs, [(t_, t_.detach().permute(0, 2, 1).contiguous()) for t_ in tcks], bias, 3, 3, 1, share_x=True, co=32)
                y.backward(dy)
            return y.detach(), xs.grad, [t_.grad[:, :4] for t_ in tcks], bias.grad
        y, dx, dws, db = run_si()
        assert y.dtype == B16 and dx.dtype == torch.float32
        with hip.option('debug_noc4', 1):
            y_p, dx_p, dws_p, db_p = run_si()
        assert not torch.equal(y, y_p), 'the Cin = 4 kernel did not run'
        assert torch.equal(dx, dx_p) and all(torch.equal(a, c) for a, c in zip(dws, dws_p))       # the backward is the same padded path
        for g in range(2):
            want = F.conv2d(x[S].cpu(), ws[g], b, 1, 1)                       # fp32 operands, bf16 result
            assert rel(y[[g * N + s_ for s_ in S]], want) <= 4e-3, ('si fwd vs torch', g)
            assert rel(y_p[[g * N + s_ for s_ in S]], F.conv2d(r16(x[S].cpu()), r16(ws[g]), b, 1, 1)) <= 4e-3
        xr = r16(x[S].cpu()).requires_grad_(True)
        wr = [r16(w).requires_grad_(True) for w in ws]
        tot = sum((F.conv2d(xr, wr[g], None, 1, 1) * dy[[g * N + s_ for s_ in S]].float().cpu()).sum() for g in range(2))
        tot.backward()
        assert rel(dx[S], xr.grad) <= 1e-2, 'si dgrad vs torch'             # two bf16 group gradients, added in bf16: three roundings
        del y, y_p, dx, dx_p, dy
        # ---- 1x1 head: bf16 trunk in, fp32 out
        xh = torch.randn((2 * N, 16, H, H), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last).to(B16)
        wh = [rnd((7, 16, 1, 1), 30 + g, 0.25) for g in range(2)]; bh = rnd((7,), 33, 0.1)
        dyh = torch.randn((2 * N, 7, H, H), device=DEV, generator=gen).contiguous(memory_format=torch.channels_last)
        dyz = torch.zeros_like(dyh)
        idx = [g * N + s_ for g in range(2) for s_ in S]
        dyz[idx] = dyh[idx]

        def run_head(dy_):
            xs = xh.detach().requires_grad_(True)
            tcks = [to_tck(w).to(DEV).requires_grad_(True) for w in wh]
            bias = bh.to(DEV).requires_grad_(True)
            with ops.mix_cache():
                y = ops.conv2d_grouped(xs, [(t_, t_.detach().permute(0, 2, 1).contiguous()) for t_ in tcks], bias, 1, 1, 0, co=7)
                y.backward(dy_)
            return y.detach(), xs.grad, [t_.grad for t_ in tcks], bias.grad
        yh, dxh, _, _ = run_head(dyh)
        assert yh.dtype == torch.float32 and dxh.dtype == B16
        _, _, dwz, dbz = run_head(dyz)
        xs_c = xh[idx].float().cpu().requires_grad_(True)
        wc = [w.clone().requires_grad_(True) for w in wh]
        outs = [F.conv2d(xs_c[2 * g:2 * g + 2], wc[g], bh) for g in range(2)]
        for g in range(2):
            assert rel(yh[idx[2 * g:2 * g + 2]], outs[g]) <= 2e-5, ('head fwd vs torch', g)       # fp32 arithmetic on the stored bf16 values
        sum((o * dyh[idx[2 * g:2 * g + 2]].cpu()).sum() for g, o in enumerate(outs)).backward()
        assert rel(dxh[idx], xs_c.grad) <= 4e-3, 'head dgrad vs torch (bf16 result)'
        for g in range(2):
            assert rel(dwz[g], to_tck(wc[g].grad)) <= 2e-5, ('head wgrad vs torch', g)
        assert rel(dbz, dyz.sum((0, 2, 3)).cpu()) <= 2e-5
    finally:
        ops.set_compute_dtype('f32')


def test_full_step_at_bench_scale_bf16(mrdis):
    """One B = 32, M = 4, 256x256 training step (adversarial loss on) under `compute_dtype: bf16` -- the BASELINE configs[2] workload
    per GPU: (a) the pipelined / packed kernels (default) vs the plain ones (wino_pipe = 0, debug_nopack = 1): identical arithmetic, so
    loss, gradient norm and the weights after Adam agree to fp32 rounding; (b) against the fp32 step on the same batch and weights: the
    difference is the bf16 rounding of activations and MFMA operands -- tolerances = 5x what was measured when the test was written
    (DESIGN.md section 4.1 records the values)."""
    m = mrdis
    l_b, p_b, g_b, w_b, names, offs = _one_step(m, 'bf16')
    l_p, p_p, g_p, w_p, _, _ = _one_step(m, 'bf16', opts=(('wino_pipe', 0), ('debug_nopack', 1)))
    assert np.isfinite(l_b) and abs(l_b - l_p) <= 1e-6 * abs(l_p), (l_b, l_p)
    assert abs(g_b - g_p) <= 1e-4 * g_p, (g_b, g_p)           # (measured 1.1e-5: wino_pipe = 0 also takes the SPADE modulation out of the convolution's epilogue)
    assert float((w_b - w_p).abs().max()) <= 4.2e-4 and float((w_b - w_p).abs().mean()) <= 1e-7
    l_f, p_f, g_f, w_f, names_f, _ = _one_step(m, 'f32')
    assert names == names_f
    rec = dict(loss_bf16=l_b, loss_f32=l_f, loss_rel=abs(l_b - l_f) / abs(l_f), gnorm_bf16=g_b, gnorm_f32=g_f, gnorm_rel=abs(g_b - g_f) / g_f,
               parts_rel={k_: abs(p_b[k_] - p_f[k_]) / (abs(p_f[k_]) + 1e-12) for k_ in p_f if abs(p_f[k_]) > 0},
               w_mean_abs_diff=float((w_b - w_f).abs().mean()))
    dump_measured('bf16_scale_measured.json', rec, mode='w')
    # measured when written (round 3): loss 6.9e-5, gradient norm 1.6e-3, recon_x / recon_x_mix 9e-7, sim_z 7e-8, latent_z 8.6e-4,
    # sim_s 1.4e-3, adv_s / adv_s_d 6.6e-5
    assert rec['loss_rel'] <= 3.5e-4, rec
    assert rec['gnorm_rel'] <= 8e-3, rec
    tol = dict(recon_x=5e-6, recon_x_mix=5e-6, sim_z=5e-6, latent_z=4.5e-3, sim_s=7e-3, adv_s=3.5e-4, adv_s_d=3.5e-4, all=3.5e-4)
    for k_, v in rec['parts_rel'].items():
        assert v <= tol[k_], (k_, rec)


@pytest.mark.parametrize('drop', ['random', 'absent3'])
def test_full_step_at_bench_scale_missing_modality(mrdis, drop):
    """BASELINE configs[3] at the benchmarked size (B = 32, M = 4, 256x256, what `bench.py --drop` times): the missing-modality step -- drop-off masks
    in every loss (model.py:3262-3266, 3319-3341, 3388), pruned decoder groups, the gated Adam step -- under the default kernel policy against the direct
    kernels only (wino = 0): loss, every loss part, the global gradient norm and the weights after Adam.  'absent3': a modality missing from the whole
    batch -- its decoder receives no gradient in the reference (torch's Adam skips `grad is None`): every one of its weights must be untouched, its
    moments zero, while every other group moves."""
    m = mrdis
    ex = {}
    l1, p1, g1, w1, names, offs = _one_step(m, 'f32', drop=drop, extra=ex)
    l0, p0, g0, w0, names0, offs0 = _one_step(m, 'f32', drop=drop, opts=(('wino', 0),))
    assert names == names0 and offs == offs0 and ex['skipped'] == 0
    assert np.isfinite(l1) and abs(l1 - l0) <= 1e-4 * abs(l0), (l1, l0)
    for k_ in p0:
        assert abs(p1[k_] - p0[k_]) <= 2e-4 * abs(p0[k_]) + 1e-7, (k_, p1[k_], p0[k_])
    assert abs(g1 - g0) <= 1e-3 * g0, (g1, g0)
    assert not torch.equal(w1, w0), 'the default policy did not change any kernel'
    # Adam's first step moves a weight by ~lr = 2e-4 whatever the gradient's size, and there are two of them (generator + discriminator optimizer):
    # a noise-level gradient may flip its sign between the two policies
    d = (w1 - w0).abs()
    assert float(d.max()) <= 8.4e-4 and float(d.mean()) <= 2e-6, (float(d.max()), float(d.mean()))
    act = ex['active']
    assert act.shape == (4,)
    if drop == 'random':
        assert act.tolist() == [1, 1, 1, 1]
    else:
        assert act.tolist() == [1, 1, 1, 0]               # (with modality 3 -- the last one -- absent the mix loss's non-advancing index reads no decoder-3 output)
    moved = (w1 - ex['w_before']).abs()
    bounds = offs + [w1.numel()]
    seen = {0: 0, 1: 0}
    for k, n in enumerate(names):
        seg = slice(bounds[k], bounds[k + 1])
        dec = [i for i in range(4) if n.startswith(f'input_decoder_list.{i}.')]
        if dec and not act[dec[0]]:
            assert float(moved[seg].max()) == 0.0 and float(ex['m_after'][seg].abs().max()) == 0.0, n     # gated off: untouched, no moment
            seen[0] += 1
        elif dec:
            seen[1] += int(float(moved[seg].max()) > 0)
    assert seen[1] > 100 and (drop == 'random' or seen[0] > 30), seen
