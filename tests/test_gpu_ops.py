"""GPU parity of every C-ABI entry point against plain fp32 torch on the CPU
(the published primitives the reference's call sites dispatch to).
Tolerance: BASELINE.json north_star -> 1e-3 relative; most checks are tighter."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def mrdis():
    import mrdis as m
    assert torch.cuda.is_available()
    m.hip.load()
    return m


def dev():
    return torch.device('cuda:0')


def cl(x):
    return x.to(dev()).contiguous(memory_format=torch.channels_last)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, want, rtol=1e-4, atol=None, what=''):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if atol is None:
        atol = rtol * float(want.abs().max()) + 1e-7
    err = (got - want).abs().max().item()
    assert torch.isfinite(got).all(), what
    assert err <= atol + rtol * float(want.abs().max()), (what, err, float(want.abs().max()))


def to_tck(w):   # (Co,Ci,kh,kw) -> [T][Ci][Co]
    Co, Ci, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw, Ci, Co).contiguous()


def to_tkc(w):   # -> [T][Co][Ci]
    Co, Ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, Co, Ci).contiguous()


# (N, Ci, Co, H, W, k, stride, pad): the path's layer geometries at reduced extents + ragged cases
CONV_CASES = [
    (2, 4, 32, 48, 40, 3, 1, 1),      # sp*.si_layers
    (2, 7, 32, 32, 64, 4, 2, 1),      # ana_enc.down_1
    (3, 7, 16, 32, 32, 3, 2, 1),      # mod_enc.conv1
    (2, 32, 64, 16, 24, 4, 2, 1),     # ana_enc.down_2
    (2, 64, 128, 10, 12, 4, 2, 1),    # down_3-like, ragged grid
    (2, 16, 32, 20, 24, 3, 2, 1),     # mod_enc.conv2
    (2, 128, 128, 5, 6, 3, 1, 1),     # sp1 gamma/beta at the 5x6 grid
    (4, 128, 64, 10, 12, 3, 1, 1),    # sp4.out
    (1, 256, 256, 10, 12, 3, 1, 1),   # ana_dec.up_4
    (2, 96, 40, 20, 24, 3, 1, 1),     # channel counts that are not tile multiples
    (2, 64, 4, 40, 48, 3, 1, 1),      # ana_dec.output (Cout = 4)
    (2, 16, 7, 32, 48, 1, 1, 0),      # decoder 1x1 out conv
    (2, 4, 16, 32, 32, 4, 2, 1),      # discriminator conv1
    (2, 64, 1, 5, 6, 3, 1, 1),        # PatchGAN head
    (1, 5, 6, 9, 11, 3, 1, 1),        # odd everything
    (3, 4, 6, 11, 13, 3, 2, 1),       # odd extents, stride 2
    (2, 32, 32, 33, 37, 3, 1, 1),     # ragged tiles
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_fwd_bwd(mrdis, case):
    N, Ci, Co, H, W, k, s, p = case
    x = rnd((N, Ci, H, W), 1).requires_grad_(True)
    w = rnd((Co, Ci, k, k), 2, 0.2).requires_grad_(True)
    b = rnd((Co,), 3, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, s, p)
    gy = rnd(tuple(y.shape), 4)
    y.backward(gy)
    hip = mrdis.hip
    got = hip.conv2d_fwd(cl(x.detach()), to_tck(w.detach()).to(dev()), b.detach().to(dev()), k, k, s, p)
    close(got, y, what='fwd')
    got_l = hip.conv2d_fwd(cl(x.detach()), to_tck(w.detach()).to(dev()), b.detach().to(dev()), k, k, s, p, lrelu=True)
    close(got_l, F.leaky_relu(y, 0.2), what='fwd+lrelu')
    dx = hip.conv2d_bwd_data(cl(gy), to_tkc(w.detach()).to(dev()), (H, W), k, k, s, p)
    close(dx, x.grad, what='dgrad')
    dw, db = hip.conv2d_bwd_weight(cl(x.detach()), cl(gy), k, k, s, p, need_bias=True)
    close(dw, to_tck(w.grad), rtol=2e-4, what='wgrad')
    close(db, b.grad, rtol=2e-4, what='dbias')


WINO_CASES = [
    (2, 32, 32, 40, 56),      # 32-cout variant (two 4-wave workgroups per CU), several tile blocks
    (1, 64, 64, 31, 45),      # odd extents: half tiles at the bottom / right edges
    (2, 128, 64, 16, 16),     # 64-cout variant, one tile block per image
    (1, 16, 48, 20, 20),      # cout tail inside a 64-wide tile
    (1, 24, 96, 18, 34),      # channel tail inside the last 8-channel chunk
    (3, 512, 128, 8, 8),      # deep reduction (64 chunks)
    (2, 64, 32, 22, 30),      # weight gradient: 64 x 32 blocks (4 waves)
    (2, 32, 64, 17, 21),      # weight gradient: 32 x 64 blocks, odd extents
    (4, 128, 128, 32, 32),    # weight gradient: several (ci, co) blocks, several tile blocks per split
    (1, 64, 64, 1, 1),        # a single pixel: every tile is partial
    (2, 64, 64, 3, 5),        # smaller than one tile block
]


@pytest.mark.parametrize('case', WINO_CASES, ids=[str(c) for c in WINO_CASES])
def test_conv_winograd_forced(mrdis, case):
    """mrdis_wino.hip (fused Winograd F(2x2,3x3)) forced on (MRDIS_WINO=2) for shapes the size policy would send to the
    direct kernel: forward (+ bias, + LeakyReLU), data gradient, and a strided (channel-slice) input view; and the two
    kernels against each other."""
    N, Ci, Co, H, W = case
    hip = mrdis.hip
    x = rnd((N, Ci, H, W), 1).requires_grad_(True)
    w = rnd((Co, Ci, 3, 3), 2, 0.2).requires_grad_(True)
    b = rnd((Co,), 3, 0.1)
    y = F.conv2d(x, w, b, 1, 1)
    gy = rnd(tuple(y.shape), 4)
    y.backward(gy)
    w_tck, w_tkc = to_tck(w.detach()).to(dev()), to_tkc(w.detach()).to(dev())
    hip.set_option('wino', 0)
    y_direct = hip.conv2d_fwd(cl(x.detach()), w_tck, b.to(dev()), 3, 3, 1, 1)
    hip.set_option('wino', 2)
    y_w = hip.conv2d_fwd(cl(x.detach()), w_tck, b.to(dev()), 3, 3, 1, 1)
    assert not torch.equal(y_w, y_direct)                      # really another kernel
    close(y_w, y, rtol=1e-4, what='winograd fwd')
    close(y_w, y_direct, rtol=1e-4, what='winograd vs direct')
    close(hip.conv2d_fwd(cl(x.detach()), w_tck, b.to(dev()), 3, 3, 1, 1, lrelu=True), F.leaky_relu(y, 0.2), rtol=1e-4, what='fwd+lrelu')
    close(hip.conv2d_bwd_data(cl(gy), w_tkc, (H, W), 3, 3, 1, 1), x.grad, rtol=1e-4, what='winograd dgrad')
    dw, db = hip.conv2d_bwd_weight(cl(x.detach()), cl(gy), 3, 3, 1, 1, need_bias=True)     # Winograd where Ci / Co are 32 / 64 multiples
    close(dw, to_tck(w.grad), rtol=2e-4, what='winograd wgrad')
    close(db, gy.sum((0, 2, 3)), rtol=2e-4, what='dbias')
    wide = cl(torch.cat([rnd((N, 8, H, W), 9), x.detach()], 1))          # channel slice: ld = Ci + 8
    close(hip.conv2d_fwd(wide[:, 8:], w_tck, None, 3, 3, 1, 1), y - b.view(1, -1, 1, 1), rtol=1e-4, what='strided view')


@pytest.mark.parametrize('case', [(2, 32, 64, 23, 37), (3, 64, 40, 50, 33), (1, 96, 16, 9, 70), (6, 128, 256, 64, 64), (2, 64, 32, 40, 96)], ids=str)
def test_bf16_pipelined_conv_bit_identical(mrdis, case):
    """bconv3_kernel (mrdis_bf16p.hip: 3x3 s1 on bf16 activations, Cin % 32 == 0, maps >= 32 wide; option wino_pipe = 1) computes the
    same products in the same order as bconv_kernel (wino_pipe = 0): forward (+ bias, LeakyReLU) and data gradient must be
    bit-identical, ragged tiles, cout tails and several units per workgroup included; and both agree with torch on bf16-rounded operands."""
    N, Ci, Co, H, W = case
    hip = mrdis.hip
    B16 = torch.bfloat16
    x = rnd((N, Ci, H, W), 1); w = rnd((Co, Ci, 3, 3), 2, 0.2); b = rnd((Co,), 3, 0.1); gy = rnd((N, Co, H, W), 4)
    xb, gyb = cl(x).to(B16), cl(gy).to(B16)
    w_tck, w_tkc = to_tck(w).to(dev()), to_tkc(w).to(dev())
    wb_f, wb_b = hip.cast_bf16(w_tkc), hip.cast_bf16(w_tck)
    out = {}
    for pipe in (0, 1):
        hip.set_option('wino_pipe', pipe)
        y = hip.conv2d_fwd(xb, w_tck, b.to(dev()), 3, 3, 1, 1, lrelu=True, w_bf16=wb_f)
        g = hip.conv2d_bwd_data(gyb, w_tkc, (H, W), 3, 3, 1, 1, w_bf16=wb_b) if Co % 32 == 0 else None
        out[pipe] = (y, g)
    assert out[1][0].dtype == B16 and torch.equal(out[0][0], out[1][0])
    if out[1][1] is not None:
        assert torch.equal(out[0][1], out[1][1])
    ref = F.leaky_relu(F.conv2d(x.to(B16).float(), w.to(B16).float(), b, 1, 1), 0.2)
    close(out[1][0].float(), ref, rtol=1.5e-2, what='bf16 pipelined fwd vs torch on bf16-rounded operands')


BCONV4_CASES = [(2, 32, 64, 23, 37), (3, 64, 40, 50, 33), (1, 96, 16, 9, 70), (6, 128, 256, 64, 64), (2, 64, 32, 40, 96), (5, 32, 72, 16, 32), (1, 64, 128, 130, 67),
                (9, 32, 16, 48, 64)]


@pytest.mark.parametrize('case', BCONV4_CASES, ids=str)
def test_bf16_lds_dma_conv_bit_identical(mrdis, case):
    """bconv4_kernel (mrdis_bf16q.hip: operand images by LDS-DMA into swizzled 64-byte rows, 16 x 32 tiles, a wave = 64 positions x 64 couts; option bconv4 = 2:
    wherever it applies) against bconv3_kernel (bconv4 = 0): the same products in the same order, so forward (+ bias, LeakyReLU) and data gradient are
    BIT-IDENTICAL -- ragged tiles in both directions, maps of a single tile row, cout tails (40, 72), <= 32 couts (the 32-cout instantiation), several
    units and chunks per workgroup; and agree with torch on the bf16-rounded operands.  hip.launch_counts() proves which kernel ran."""
    N, Ci, Co, H, W = case
    hip = mrdis.hip
    B16 = torch.bfloat16
    x = rnd((N, Ci, H, W), 1); w = rnd((Co, Ci, 3, 3), 2, 0.2); b = rnd((Co,), 3, 0.1); gy = rnd((N, Co, H, W), 4)
    xb, gyb = cl(x).to(B16), cl(gy).to(B16)
    w_tck, w_tkc = to_tck(w).to(dev()), to_tkc(w).to(dev())
    wb_f, wb_b = hip.cast_bf16(w_tkc), hip.cast_bf16(w_tck)
    out = {}
    for mode in (0, 2):                                 # bconv3 | bconv4 wherever it applies
        with hip.option('bconv4', mode):
            hip.launch_counts(reset=True)
            y = hip.conv2d_fwd(xb, w_tck, b.to(dev()), 3, 3, 1, 1, lrelu=True, w_bf16=wb_f)
            y0 = hip.conv2d_fwd(xb, w_tck, None, 3, 3, 1, 1, w_bf16=wb_f)
            g = hip.conv2d_bwd_data(gyb, w_tkc, (H, W), 3, 3, 1, 1, w_bf16=wb_b) if Co % 32 == 0 else None
            c = hip.launch_counts()
        assert (c['bconv4'] > 0) == (mode != 0) and (c['bconv3'] > 0) == (mode == 0), (mode, c)
        out[mode] = (y, y0, g)
    for mode in (2,):
        assert out[mode][0].dtype == B16 and torch.equal(out[0][0], out[mode][0]) and torch.equal(out[0][1], out[mode][1]), mode
        if out[mode][2] is not None:
            assert torch.equal(out[0][2], out[mode][2]), mode
    ref = F.leaky_relu(F.conv2d(x.to(B16).float(), w.to(B16).float(), b, 1, 1), 0.2)
    close(out[2][0].float(), ref, rtol=1.5e-2, what='bf16 LDS-DMA fwd vs torch on bf16-rounded operands')
    # a channel-slice input view (ld > Cin) whose last pixel ends exactly at the descriptor's record count, and an output slice of a wider buffer
    if Co % 8 == 0:
        wide_in = cl(torch.cat([rnd((N, 8, H, W), 9), x], 1)).to(B16)
        wide_out = hip.empty_nhwc(N, Co + 8, H, W, dev(), B16); wide_out.zero_()
        with hip.option('bconv4', 2):
            hip.conv2d_fwd(wide_in[:, 8:], w_tck, b.to(dev()), 3, 3, 1, 1, lrelu=True, w_bf16=wb_f, out=wide_out[:, 8:])
        assert torch.equal(wide_out[:, 8:], out[0][0]) and float(wide_out[:, :8].float().abs().max()) == 0.0


@pytest.mark.parametrize('case', [(8, 32, 32, 64, 96), (3, 64, 64, 50, 72), (5, 128, 128, 33, 47), (2, 32, 48, 64, 80), (4, 32, 16, 16, 32)], ids=str)
def test_gb_spade_fused_epilogue_bf16_lds_dma(mrdis, case):
    """the SPADE-fused form of bconv4_kernel against bconv3_kernel<2, 0, true>: mix and gamma bit-identical (ragged tiles, a channel count that is not a multiple of 32)."""
    N, Ci, C, H, W = case
    hip = mrdis.hip
    x = cl(rnd((N, Ci, H, W), 1)).to(torch.bfloat16); z = cl(rnd((N, C, H, W), 2)).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last); z = z.contiguous(memory_format=torch.channels_last)
    w = rnd((2 * C, Ci, 3, 3), 3, 0.2); b = rnd((2 * C,), 4, 0.1).to(dev())
    w_tck = to_tck(w).to(dev()); wb = hip.cast_bf16(to_tkc(w).to(dev()))
    res = {}
    for mode in (0, 2):
        with hip.option('bconv4', mode):
            hip.launch_counts(reset=True)
            res[mode] = hip.gb_spade_fwd(x, w_tck, b, z, 1e-5, w_bf16=wb)
            c = hip.launch_counts()
        assert res[mode] is not None and (c['bconv4_spade'] > 0) == (mode != 0) and (c['bconv3_spade'] > 0) == (mode == 0), (mode, c)
    for mode in (2,):
        for a, b_ in zip(res[0], res[mode]):
            assert torch.equal(a, b_), mode


def test_to_device_mailbox_ring(mrdis):
    """ops.to_device for small tensors: pinned ring slot + copy kernel (no copy engine).  More transfers than ring slots, mixed dtypes
    and sizes, all checked only at the end (the GPU consumes the slots while the host keeps refilling them)."""
    ops = mrdis.ops
    g = torch.Generator().manual_seed(5)
    sent, got = [], []
    filler = torch.zeros(1 << 22, device=dev())
    for k in range(3 * mrdis.hip._Mailbox.NSLOT + 7):
        if k % 3 == 0:
            t = torch.randn(32, 16, generator=g)
        elif k % 3 == 1:
            t = torch.randint(0, 1 << 40, (2, 1 + k % 5), generator=g, dtype=torch.long)
        else:
            t = torch.randn(1 + (k * 37) % 4000, generator=g)
        filler.add_(1.0)                               # keeps the GPU behind the host
        sent.append(t); got.append(ops.to_device(t, dev()))
    torch.cuda.synchronize()
    for a, b in zip(sent, got):
        assert b.device.type == 'cuda' and b.dtype == a.dtype and torch.equal(a, b.cpu())
    big = torch.randn(1 << 16)                         # 256 KB: beyond a slot -> pinned-cache path
    assert torch.equal(ops.to_device(big, dev()).cpu(), big)


@pytest.mark.parametrize('case', [(8, 32, 32, 64, 96), (3, 64, 64, 50, 72), (5, 128, 128, 33, 47), (2, 32, 48, 64, 80), (2, 16, 128, 8, 8)], ids=str)
def test_gb_spade_fused_epilogue(mrdis, case):
    """mrdis_conv2d_fwd_spade (the gamma | beta convolution with the InstanceNorm modulation in the Winograd kernel's epilogue) against the
    two-step form (convolution, then instnorm_spade): mix and gamma, ragged tile blocks, a channel count that is not a multiple of 32
    (48: half-filled last workgroup), and a geometry the fused kernel declines (-> None)."""
    N, Ci, C, H, W = case
    hip = mrdis.hip
    x = cl(rnd((N, Ci, H, W), 1)); z = cl(rnd((N, C, H, W), 2))
    w = rnd((2 * C, Ci, 3, 3), 3, 0.2); b = rnd((2 * C,), 4, 0.1).to(dev())
    w_tck = to_tck(w).to(dev())
    gb = hip.conv2d_fwd(x, w_tck, b, 3, 3, 1, 1)
    mix_ref, mean_ref, rstd_ref = hip.instnorm_spade_fwd(z, gb[:, :C], gb[:, C:], 1e-5)
    if H < 16:
        assert hip.gb_spade_fwd(x, w_tck, b, z, 1e-5) is None          # too few tile blocks for the Winograd policy: the caller falls back
        return
    hip.set_option('wino', 2)                           # Winograd wherever the kernel applies (the size policy is tested at scale)
    res = hip.gb_spade_fwd(x, w_tck, b, z, 1e-5)
    assert res is not None
    mix, gamma, mean, rstd = res
    close(gamma, gb[:, :C].cpu(), rtol=2e-5, what='gamma')
    close(mix, mix_ref.cpu(), rtol=2e-5, what='mix')
    close(mean, mean_ref.cpu(), rtol=1e-6, what='mean'); close(rstd, rstd_ref.cpu(), rtol=1e-6, what='rstd')
    zr = z.float().cpu().permute(0, 2, 3, 1); gr = F.conv2d(x.cpu(), w, b.cpu(), 1, 1)
    ref = F.instance_norm(z.cpu(), eps=1e-5) * (1 + gr[:, :C]) + gr[:, C:]
    close(mix, ref, rtol=1e-4, what='mix vs torch')


@pytest.mark.parametrize('case', [(8, 32, 32, 64, 96), (3, 64, 64, 50, 72), (5, 128, 128, 33, 47), (2, 32, 48, 64, 80), (2, 32, 128, 8, 8)], ids=str)
def test_gb_spade_fused_epilogue_bf16(mrdis, case):
    """the bf16 form of mrdis_conv2d_fwd_spade (SPADE epilogue of the pipelined bf16 kernel) against the two-step bf16 path
    (bf16 convolution -> bf16 gamma | beta, then instnorm_spade): the same accumulation order and the same roundings, so mix and
    gamma must be bit-identical; a map narrower than the kernel's 32-pixel tile is declined (-> None)."""
    N, Ci, C, H, W = case
    hip = mrdis.hip
    x = cl(rnd((N, Ci, H, W), 1)).to(torch.bfloat16); z = cl(rnd((N, C, H, W), 2)).to(torch.bfloat16)
    x = x.contiguous(memory_format=torch.channels_last); z = z.contiguous(memory_format=torch.channels_last)
    w = rnd((2 * C, Ci, 3, 3), 3, 0.2); b = rnd((2 * C,), 4, 0.1).to(dev())
    w_tck = to_tck(w).to(dev()); wb = hip.cast_bf16(to_tkc(w).to(dev()))
    gb = hip.conv2d_fwd(x, w_tck, b, 3, 3, 1, 1, w_bf16=wb)
    assert gb.dtype == torch.bfloat16
    mix_ref, mean_ref, rstd_ref = hip.instnorm_spade_fwd(z, gb[:, :C], gb[:, C:], 1e-5)
    res = hip.gb_spade_fwd(x, w_tck, b, z, 1e-5, w_bf16=wb)
    if W < 32:
        assert res is None
        return
    assert res is not None
    mix, gamma, mean, rstd = res
    assert mix.dtype == torch.bfloat16 and gamma.dtype == torch.bfloat16
    assert torch.equal(mean, mean_ref) and torch.equal(rstd, rstd_ref)
    assert torch.equal(gamma, gb[:, :C]), 'gamma differs from the two-step path'
    assert torch.equal(mix, mix_ref), 'mix differs from the two-step path'
    gr = F.conv2d(x.float().cpu(), w, b.cpu(), 1, 1)
    ref = F.instance_norm(z.float().cpu(), eps=1e-5) * (1 + gr[:, :C]) + gr[:, C:]
    err = (mix.float().cpu() - ref).norm() / ref.norm()
    assert err < 1e-2, err


PIPE_CASES = [
    (8, 64, 128, 96, 80),     # 480 blocks on <= 256 persistent workgroups: every workgroup walks several blocks, both cout tiles
    (3, 36, 72, 50, 18),      # channel tail in the last chunk, cout tail in the second 64-wide tile, partial tile blocks
    (2, 8, 40, 23, 37),       # one chunk per block: the pipeline turns over at every iteration
    (1, 256, 64, 33, 47),     # 32 chunks per block, odd extents
    (5, 128, 256, 16, 16),    # 4 cout tiles, one tile block per image
]


@pytest.mark.parametrize('case', PIPE_CASES, ids=[str(c) for c in PIPE_CASES])
def test_winograd_pipelined_kernel(mrdis, case):
    """mrdis_wino2.hip (option wino_pipe = 1, the default for Cout > 32) against torch and against the phase-by-phase
    kernel it replaces (wino_pipe = 0): forward with bias + LeakyReLU, data gradient (flipped taps, no bias), and a
    channel-slice input view whose last pixel ends exactly at the buffer descriptor's record count."""
    N, Ci, Co, H, W = case
    hip = mrdis.hip
    x = rnd((N, Ci, H, W), 1).requires_grad_(True)
    w = rnd((Co, Ci, 3, 3), 2, 0.2).requires_grad_(True)
    b = rnd((Co,), 3, 0.1)
    y = F.conv2d(x, w, b, 1, 1)
    gy = rnd(tuple(y.shape), 4)
    y.backward(gy)
    w_tck, w_tkc = to_tck(w.detach()).to(dev()), to_tkc(w.detach()).to(dev())
    hip.set_option('wino', 2)
    out = {}
    for pipe in (0, 1):
        hip.set_option('wino_pipe', pipe)
        out[pipe] = (hip.conv2d_fwd(cl(x.detach()), w_tck, b.to(dev()), 3, 3, 1, 1, lrelu=True),
                     hip.conv2d_bwd_data(cl(gy), w_tkc, (H, W), 3, 3, 1, 1))
    close(out[1][0], F.leaky_relu(y, 0.2), rtol=1e-4, what='pipelined fwd')
    close(out[1][1], x.grad, rtol=1e-4, what='pipelined dgrad')
    close(out[1][0], out[0][0], rtol=2e-6, what='pipelined vs phase kernel, fwd')          # same arithmetic, same order
    close(out[1][1], out[0][1], rtol=2e-6, what='pipelined vs phase kernel, dgrad')
    wide = cl(torch.cat([rnd((N, 8, H, W), 9), x.detach()], 1))          # channel slice: ld = Ci + 8, view ends at the allocation's end
    close(hip.conv2d_fwd(wide[:, 8:], w_tck, None, 3, 3, 1, 1), y - b.view(1, -1, 1, 1), rtol=1e-4, what='strided view')


@pytest.mark.parametrize('case', [(2, 64, 64, 23, 37), (3, 128, 64, 50, 18), (4, 128, 128, 32, 32), (9, 64, 192, 40, 72), (1, 64, 64, 2, 3)],
                         ids=str)
def test_winograd_pipelined_wgrad(mrdis, case):
    """wino_wgrad2_kernel (mrdis_wino2.hip; 64 x 64 (ci, co) blocks, option wino_pipe = 1) against torch and against the
    phase-by-phase kernel: weight and bias gradient, odd extents (partial 2 x 4 tile blocks), more tile blocks than splits
    (every workgroup walks several), a map smaller than one tile block."""
    N, Ci, Co, H, W = case
    hip = mrdis.hip
    x = rnd((N, Ci, H, W), 1); gy = rnd((N, Co, H, W), 2)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    F.conv2d(x, w, b, 1, 1).backward(gy)
    hip.set_option('wino', 2)
    out = {}
    for pipe in (0, 1):
        hip.set_option('wino_pipe', pipe)
        out[pipe] = hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True)
    close(out[1][0], to_tck(w.grad), rtol=2e-4, what='pipelined wgrad')
    close(out[1][1], b.grad, rtol=2e-4, what='pipelined dbias')
    close(out[1][0], out[0][0], rtol=1e-5, what='pipelined vs phase kernel')
    sink = torch.ones(Co, device=dev())
    hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True, bias_sink=sink)
    close(sink - 1, b.grad, rtol=2e-4, what='dbias accumulated into a sink')


@pytest.mark.parametrize('N,Ci,Co,H,W', [(2, 32, 16, 240, 232), (1, 16, 8, 321, 333), (3, 48, 12, 200, 180), (2, 32, 16, 243, 251), (5, 32, 16, 129, 160)])
def test_wgrad_narrow_cout(mrdis, N, Ci, Co, H, W):
    """mrdis_wgrad16.hip: weight / bias gradient of 3x3 s1 layers with 8..16 couts on large maps (sp6.out), 16-channel input
    slices, ragged boxes at the right / bottom edges, a cout count that is not 16."""
    x = rnd((N, Ci, H, W), 1); gy = rnd((N, Co, H, W), 2)
    w = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    b = torch.zeros(Co, requires_grad=True)
    F.conv2d(x, w, b, 1, 1).backward(gy)
    dw, db = mrdis.hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True)
    close(dw, to_tck(w.grad), rtol=2e-4, what='wgrad16')
    close(db, b.grad, rtol=2e-4, what='dbias')
    if Ci == 32 and Co == 16:
        # the six-product form (option split6: wgrad16_split6_kernel, taken by default) against the fp32 MFMA form and a float64 reference
        hip = mrdis.hip
        with hip.option('split6', 0):
            dw32, db32 = hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True)
        with hip.option('split6', 6):
            dw6, db6 = hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True)
        w64 = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double(), w64, None, 1, 1).backward(gy.double())
        ref = to_tck(w64.grad); sc = float(ref.abs().max())
        e32, e6 = float((dw32.cpu().double() - ref).abs().max()) / sc, float((dw6.cpu().double() - ref).abs().max()) / sc
        assert e6 <= 2.0 * e32 + 1e-7, ('split6 wgrad16', e6, e32)
        close(dw6, dw32, rtol=2e-6, what='split6 vs fp32 MFMA (wgrad16)'); close(db6, db32, rtol=2e-6, what='split6 bias gradient')
        assert not torch.equal(dw6, dw32), 'the six-product kernel did not run'


@pytest.mark.parametrize('wino', ['1', '2'])
def test_conv_random_shapes(mrdis, wino):
    """Seeded sweep over the dispatcher (direct MFMA / narrow / Cin = 4 / thin / Winograd / narrow-cout weight gradient):
    random channel counts (tile multiples and not), extents down to 1x1, kernels 1 / 3 / 4, strides 1 / 2 -- forward,
    data gradient, weight and bias gradient against torch fp32 on the CPU."""
    mrdis.hip.set_option('wino', int(wino))
    g = np.random.RandomState(1234 + int(wino))
    hip = mrdis.hip
    chans = [1, 3, 4, 7, 8, 12, 16, 24, 32, 40, 48, 64, 96, 128]
    for case in range(36):
        k = int(g.choice([1, 3, 3, 3, 4]))
        stride = int(g.choice([1, 1, 2]))
        pad = 0 if k == 1 else 1
        if k == 1:
            stride = 1              # 1x1 stride 2 is not a geometry of the reference; its data gradient reports MRDIS_EUNSUPPORTED
        N = int(g.randint(1, 4)); Ci = int(g.choice(chans)); Co = int(g.choice(chans))
        H = int(g.randint(k if pad == 0 else max(1, k - 2), 48)); W = int(g.randint(k if pad == 0 else max(1, k - 2), 60))
        if (H + 2 * pad - k) // stride + 1 <= 0 or (W + 2 * pad - k) // stride + 1 <= 0:
            continue
        x = rnd((N, Ci, H, W), 10 + case).requires_grad_(True)
        w = rnd((Co, Ci, k, k), 100 + case, 0.2).requires_grad_(True)
        b = rnd((Co,), 200 + case, 0.1).requires_grad_(True)
        y = F.conv2d(x, w, b, stride, pad)
        gy = rnd(tuple(y.shape), 300 + case)
        y.backward(gy)
        tag = f'case {case}: N{N} {Ci}->{Co} {H}x{W} k{k} s{stride}'
        close(hip.conv2d_fwd(cl(x.detach()), to_tck(w.detach()).to(dev()), b.detach().to(dev()), k, k, stride, pad), y, rtol=2e-4, what=tag + ' fwd')
        close(hip.conv2d_bwd_data(cl(gy), to_tkc(w.detach()).to(dev()), (H, W), k, k, stride, pad), x.grad, rtol=2e-4, what=tag + ' dgrad')
        dw, db = hip.conv2d_bwd_weight(cl(x.detach()), cl(gy), k, k, stride, pad, need_bias=True)
        close(dw, to_tck(w.grad), rtol=3e-4, what=tag + ' wgrad')
        close(db, b.grad, rtol=3e-4, what=tag + ' dbias')


def test_conv_c4_persistent_pipeline(mrdis):
    """Cin = 4 direct kernel with enough strips (> 2 per resident wave) to run its steady-state
    register pipeline, ragged right edge included; with bias and fused LeakyReLU."""
    hip = mrdis.hip
    for (N, H, W, Co) in [(16, 128, 136, 32), (12, 120, 240, 64), (3, 250, 250, 128)]:
        x = rnd((N, 4, H, W), 50); w = rnd((Co, 4, 3, 3), 51, 0.2); b = rnd((Co,), 52, 0.1)
        want = F.conv2d(x, w, b, 1, 1)
        got = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1)
        close(got, want, what=f'c4 {N}x{H}x{W}->{Co}')
        gotl = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1, lrelu=True)
        close(gotl, F.leaky_relu(want, 0.2), what='c4 lrelu')
        # option split6: six bf16 products per fp32 product on the bf16 matrix pipe (c4conv_split6_kernel) -- at the fp32 kernel's own level against float64
        want64 = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
        with hip.option('split6', 4):                   # (4: every width; the default 1 takes <= 32 couts)
            got6 = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1)
            got6l = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1, lrelu=True)
        sc = float(want64.abs().max())
        e32, e6 = float((got.cpu().double() - want64).abs().max()) / sc, float((got6.cpu().double() - want64).abs().max()) / sc
        assert e6 <= 2.0 * e32 + 1e-7, ('split6 c4 forward', e6, e32)
        close(got6, got, rtol=2e-6, what='split6 vs fp32 MFMA'); close(got6l, gotl, rtol=2e-6, what='split6 vs fp32 MFMA, lrelu')
        # option c4_grid: the persistent grid (workgroups per CU) only changes which workgroup takes which strips: bit-identical results
        for grid in (1, 2, 7):
            with hip.option('c4_grid', grid):
                gg = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1)
                assert hip.get_option('debug_c4_blocks') > 0
            assert torch.equal(gg, got), ('c4_grid', grid)


@pytest.mark.parametrize('N,Ci,Co,H,W,k,st', [(24, 32, 32, 128, 144, 3, 1), (8, 64, 96, 72, 80, 3, 1), (16, 32, 64, 64, 64, 4, 2),
                                              (12, 32, 16, 96, 96, 3, 1), (6, 64, 4, 128, 128, 3, 1),
                                              (16, 4, 64, 96, 96, 3, 1), (8, 7, 32, 128, 128, 4, 2), (8, 16, 7, 128, 96, 1, 1),
                                              (8, 4, 128, 64, 64, 3, 1)])
@pytest.mark.parametrize('wino', [0, 1])
def test_wgrad_dma_pipeline(mrdis, N, Ci, Co, H, W, k, st, wino):
    """weight gradient with several position tiles per workgroup: exercises the double-buffered LDS-DMA
    steady state (and its ragged last tiles), bias column sums included.  wino=0 pins the direct kernels
    (`wgrad_dma_kernel` and friends: the fallback of every big layer); wino=1 is the default policy, which sends
    the 32/64-blocked 3x3 cases to `wino_wgrad_kernel`."""
    hip = mrdis.hip
    hip.set_option('wino', wino)
    x = rnd((N, Ci, H, W), 60).requires_grad_(False)
    w = rnd((Co, Ci, k, k), 61, 0.1).requires_grad_(True); b = rnd((Co,), 62, 0.1).requires_grad_(True)
    pad = 0 if k == 1 else 1
    y = F.conv2d(x, w, b, st, pad)
    gy = rnd(tuple(y.shape), 63); y.backward(gy)
    dw, db = hip.conv2d_bwd_weight(cl(x), cl(gy), k, k, st, pad, need_bias=True)
    close(dw, to_tck(w.grad), rtol=3e-4, what='wgrad dma'); close(db, b.grad, rtol=3e-4, what='dbias dma')


S2_CASES = [
    (8, 7, 32, 128, 128, 4, 1),     # ana_enc.down_1 geometry, modality 1 of a 28-channel batch tensor: ldx = 28, base 28 bytes into a pixel
    (8, 7, 16, 128, 128, 3, 3),     # mod_enc.conv1 geometry, modality 3
    (5, 7, 32, 66, 160, 3, 0),      # odd row count per workgroup, 3x3 with 32 couts
    (4, 5, 16, 96, 256, 4, 2),      # Cin = 5: several all-zero (tap, ci) rows in the last tiles; the widest row the kernel stages
    (16, 3, 32, 64, 96, 4, 0),      # Cin = 3
    (12, 4, 16, 64, 256, 4, 1),     # the discriminator's first layer (4 -> 16, 4x4): 64 gradient rows = 4 of the 7 row tiles, 8 output rows per workgroup
    (6, 3, 16, 128, 128, 3, 0),     # Cin = 3, 3x3: 27 rows = 2 tiles
]


@pytest.mark.parametrize('case', S2_CASES, ids=str)
def test_wgrad_stride2_first_layers(mrdis, case):
    """mrdis_wgrad_s2.hip (Cin <= 7, stride 2, 3x3 / 4x4 taps, 16 / 32 couts) on a channel-slice view of a wider NHWC batch tensor
    (the step hands the first layers `inputs[:, 7 i : 7 i + 7]`), against torch; bias gradient both fresh and added into a sink;
    option now16 = 1 (the generic split-K kernel) must agree."""
    N, Ci, Co, H, W, k, mod = case
    hip = mrdis.hip
    full = rnd((N, 4 * Ci, H, W), 70)
    x = full[:, mod * Ci:(mod + 1) * Ci]
    w = rnd((Co, Ci, k, k), 71, 0.1).requires_grad_(True); b = rnd((Co,), 72, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, 2, 1)
    gy = rnd(tuple(y.shape), 73); y.backward(gy)
    xv = cl(full)[:, mod * Ci:(mod + 1) * Ci]
    assert hip.nhwc(xv)[1] == 4 * Ci
    dw, db = hip.conv2d_bwd_weight(xv, cl(gy), k, k, 2, 1, need_bias=True)
    close(dw, to_tck(w.grad), rtol=3e-4, what='wgrad s2'); close(db, b.grad, rtol=3e-4, what='dbias s2')
    sink = torch.full((Co,), 2.0, device=dev())
    dw2, none = hip.conv2d_bwd_weight(xv, cl(gy), k, k, 2, 1, need_bias=True, bias_sink=sink)
    assert none is None and torch.equal(dw2, dw)
    close(sink, b.grad + 2.0, rtol=3e-4, what='dbias sink')
    # forward of the same layers (conv_s2_fwd_kernel): bias, LeakyReLU, output into a channel slice
    w_tck, bd = to_tck(w.detach()).to(dev()), b.detach().to(dev())
    yy = hip.conv2d_fwd(xv, w_tck, bd, k, k, 2, 1)
    close(yy, y.detach(), rtol=2e-5, what='fwd s2')
    out = hip.empty_nhwc(N, Co + 16, H // 2, W // 2, dev()); out.fill_(3.0)
    hip.conv2d_fwd(xv, w_tck, bd, k, k, 2, 1, lrelu=True, out=out[:, 16:])
    close(out[:, 16:], F.leaky_relu(y.detach(), 0.2), rtol=2e-5, what='fwd s2 lrelu into a slice')
    assert bool((out[:, :16] == 3.0).all())
    # data gradient of the same layers (dgrad_s2_kernel: window-free GEMM + parity gather), into a dense tensor and into a channel slice
    want_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), w.detach(), gy, 2, 1)
    w_tkc = to_tkc(w.detach()).to(dev())
    dx = hip.conv2d_bwd_data(cl(gy), w_tkc, (H, W), k, k, 2, 1)
    close(dx, want_dx, rtol=2e-5, what='dgrad s2')
    hip.set_option('debug_now16', 1)
    try:
        dw3, db3 = hip.conv2d_bwd_weight(xv, cl(gy), k, k, 2, 1, need_bias=True)
        y3 = hip.conv2d_fwd(xv, w_tck, bd, k, k, 2, 1)
        dx3 = hip.conv2d_bwd_data(cl(gy), w_tkc, (H, W), k, k, 2, 1)
    finally:
        hip.set_option('debug_now16', 0)
    close(dw3, dw.cpu(), rtol=3e-4, what='generic vs s2'); close(db3, db.cpu(), rtol=3e-4, what='generic vs s2 bias')
    close(y3, yy.cpu(), rtol=2e-5, what='generic vs s2 forward'); close(dx3, dx.cpu(), rtol=2e-5, what='generic vs s2 dgrad')


@pytest.mark.parametrize('case', [(8, 7, 128, 96), (3, 8, 100, 75), (5, 4, 64, 80), (2, 1, 96, 96), (32, 7, 64, 64)], ids=str)
def test_pointwise_head_kernels(mrdis, case):
    """mrdis_pointwise.hip (the 1x1 16 -> <= 8 channel decoder head as streaming kernels): forward with bias (+ LeakyReLU), data gradient
    and weight / bias gradient against torch, with the 16-channel side a channel slice of a wider tensor (ld = 32) and a pixel count
    that is not a multiple of the grid stride; option debug_now16 = 1 (the generic tile kernels) must agree."""
    N, Co, H, W = case
    hip = mrdis.hip
    wide = rnd((N, 32, H, W), 80)
    x = wide[:, 16:].clone().requires_grad_(True)
    w = rnd((Co, 16, 1, 1), 81, 0.3).requires_grad_(True); b = rnd((Co,), 82, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b)
    gy = rnd(tuple(y.shape), 83); y.backward(gy)
    xv = cl(wide)[:, 16:]
    w_tck, w_tkc = to_tck(w.detach()).to(dev()), to_tkc(w.detach()).to(dev())
    got = {}
    for now16 in (0, 1):
        hip.set_option('debug_now16', now16)
        try:
            yy = hip.conv2d_fwd(xv, w_tck, b.detach().to(dev()), 1, 1, 1, 0)
            yl = hip.conv2d_fwd(xv, w_tck, b.detach().to(dev()), 1, 1, 1, 0, lrelu=True)
            dx = hip.conv2d_bwd_data(cl(gy), w_tkc, (H, W), 1, 1, 1, 0)
            dw, db = hip.conv2d_bwd_weight(xv, cl(gy), 1, 1, 1, 0, need_bias=True)
        finally:
            hip.set_option('debug_now16', 0)
        close(yy, y.detach(), rtol=2e-5, what=f'pw fwd now16={now16}'); close(yl, F.leaky_relu(y.detach(), 0.2), rtol=2e-5, what='pw fwd lrelu')
        close(dx, x.grad, rtol=2e-5, what=f'pw dgrad now16={now16}')
        close(dw, to_tck(w.grad), rtol=3e-4, what=f'pw wgrad now16={now16}'); close(db, b.grad, rtol=3e-4, what='pw dbias')
        got[now16] = (yy.cpu(), dx.cpu(), dw.cpu())
    for a, c in zip(got[0], got[1]):
        close(a, c, rtol=3e-4, what='streaming vs tile kernels')


@pytest.mark.parametrize('case', [(4, 32, 128, 160), (3, 16, 150, 150), (2, 32, 203, 171), (9, 32, 96, 96)], ids=str)
def test_conv3x3_16_couts_register_filter(mrdis, case):
    """mrdis_c16.hip (3x3 s1 p1, 16 output channels, 16 / 32 input channels: sp6.out): filter in registers, persistent 8 x 32 tiles.
    Against torch with bias (+ LeakyReLU), ragged tiles in both directions, input and output both channel slices of wider tensors;
    option debug_now16 = 1 (tapconv16_kernel) must agree."""
    N, Ci, H, W = case
    hip = mrdis.hip
    wide = rnd((N, Ci + 8, H, W), 84)
    x = wide[:, 4:4 + Ci]
    w = rnd((16, Ci, 3, 3), 85, 0.15); b = rnd((16,), 86, 0.1)
    want = F.conv2d(x, w, b, 1, 1)
    xv = cl(wide)[:, 4:4 + Ci]
    w_tck, bd = to_tck(w).to(dev()), b.to(dev())
    got = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
    close(got, want, rtol=2e-5, what='c16 fwd')
    out = hip.empty_nhwc(N, 32, H, W, dev())
    out.fill_(7.0)
    hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1, lrelu=True, out=out[:, 16:])
    close(out[:, 16:], F.leaky_relu(want, 0.2), rtol=2e-5, what='c16 fwd lrelu into a slice')
    assert bool((out[:, :16] == 7.0).all()), 'wrote outside its 16 channels'
    hip.set_option('debug_now16', 1)
    try:
        ref = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
    finally:
        hip.set_option('debug_now16', 0)
    close(ref, got.cpu(), rtol=2e-5, what='tile kernel vs c16')
    # the six-product form (option split6, Ci = 32: conv3x3_c16_split6_kernel) against the fp32 MFMA form (split6 = 0) and a float64 reference
    with hip.option('split6', 0):
        got32 = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
    with hip.option('split6', 5):
        got6 = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
    want64 = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    sc = float(want64.abs().max())
    e32, e6 = float((got32.cpu().double() - want64).abs().max()) / sc, float((got6.cpu().double() - want64).abs().max()) / sc
    assert e6 <= 2.0 * e32 + 1e-7, ('split6 c16 forward', e6, e32)
    close(got6, got32, rtol=2e-6, what='split6 vs fp32 MFMA (c16)')
    if Ci == 32:
        assert not torch.equal(got6, got32), 'the six-product kernel did not run'


@pytest.mark.parametrize('case', [(2, 64, 256, 256), (3, 32, 70, 256), (5, 64, 128, 128), (9, 32, 100, 128), (17, 64, 64, 64), (20, 32, 61, 64)], ids=str)
def test_conv3x3_4_couts_gemm_gather(mrdis, case):
    """mrdis_co4.hip (3x3 s1 p1, 4 output channels, 32 / 64 input channels: ana_dec.output forward, the si_layers data gradients):
    window-free MFMA GEMM Z[tap][pixel][co] + fixed-order tap gather while streaming down the rows.  Forward with bias (+ LeakyReLU)
    and the data gradient of a 4 -> C layer against torch; row counts that do not divide into the workgroups' row runs; the input a
    channel slice of a wider tensor; option debug_now16 = 1 (the packed-FMA tile kernel) must agree."""
    N, C, H, W = case
    hip = mrdis.hip
    wide = rnd((N, C + 16, H, W), 87)
    x = wide[:, 8:8 + C]
    w = rnd((4, C, 3, 3), 88, 0.15); b = rnd((4,), 89, 0.1)
    want = F.conv2d(x, w, b, 1, 1)
    xv = cl(wide)[:, 8:8 + C]
    w_tck, bd = to_tck(w).to(dev()), b.to(dev())
    got = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
    close(got, want, rtol=2e-5, what='co4 fwd')
    gotl = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1, lrelu=True)
    close(gotl, F.leaky_relu(want, 0.2), rtol=2e-5, what='co4 fwd lrelu')
    # data gradient of a 4 -> C layer: dx (N, 4, H, W) from dy = the C-channel tensor
    w2 = rnd((C, 4, 3, 3), 90, 0.15)
    want_dx = torch.nn.grad.conv2d_input((N, 4, H, W), w2, x.contiguous(), 1, 1)
    dx = hip.conv2d_bwd_data(xv, to_tkc(w2).to(dev()), (H, W), 3, 3, 1, 1)
    close(dx, want_dx, rtol=2e-5, what='co4 dgrad')
    hip.set_option('debug_now16', 1)
    try:
        ref = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
        ref_dx = hip.conv2d_bwd_data(xv, to_tkc(w2).to(dev()), (H, W), 3, 3, 1, 1)
    finally:
        hip.set_option('debug_now16', 0)
    close(ref, got.cpu(), rtol=2e-5, what='tile kernel vs co4'); close(ref_dx, dx.cpu(), rtol=2e-5, what='tile kernel vs co4 dgrad')
    # option split6: the same products from three bf16 terms per operand on the bf16 matrix pipe (six of the nine, the rest below 2^-23 of a product) --
    # at the exact fp32 kernel's own level against a float64 reference: its error may not exceed twice the fp32 kernel's (+ 1e-7 of the scale)
    want64 = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    with hip.option('split6', 1):
        got6 = hip.conv2d_fwd(xv, w_tck, bd, 3, 3, 1, 1)
        dx6 = hip.conv2d_bwd_data(xv, to_tkc(w2).to(dev()), (H, W), 3, 3, 1, 1)
    sc = float(want64.abs().max())
    e32, e6 = float((got.cpu().double() - want64).abs().max()) / sc, float((got6.cpu().double() - want64).abs().max()) / sc
    assert e6 <= 2.0 * e32 + 1e-7, ('split6 forward', e6, e32)
    close(got6, got, rtol=2e-6, what='split6 vs fp32 MFMA, forward'); close(dx6, dx, rtol=2e-6, what='split6 vs fp32 MFMA, data gradient')


@pytest.mark.parametrize('case', [(2, 240, 232), (1, 321, 333), (5, 129, 160), (3, 256, 256)], ids=str)
def test_bf16_weight_gradient_32_to_16(mrdis, case):
    """wgrad16_bf16_kernel (mrdis_wgrad16.hip): sp6.out's weight + bias gradient on bf16 views (MRDIS_DT_BF16) -- K = 32 positions per
    v_mfma_f32_16x16x32_bf16, transposing LDS reads, the tap shift on the 16-cout operand.  Exact products of the bf16 operands, fp32 sums: against torch
    fp32 on the same bf16-valued operands to fp32 rounding, and against the generic bf16 kernel (debug_mode 3030: bwgrad3 / bwgrad), which it replaces."""
    N, H, W = case
    hip = mrdis.hip
    B16 = torch.bfloat16
    x = rnd((N, 32, H, W), 131).bfloat16(); dy = rnd((N, 16, H, W), 132).bfloat16()
    w0 = torch.zeros(16, 32, 3, 3, requires_grad=True); b0 = torch.zeros(16, requires_grad=True)
    F.conv2d(x.float(), w0, b0, 1, 1).backward(dy.float())
    xd, dyd = cl(x.float()).to(B16), cl(dy.float()).to(B16)
    dw, db = hip.conv2d_bwd_weight(xd, dyd, 3, 3, 1, 1, need_bias=True)
    close(dw, to_tck(w0.grad), rtol=2e-5, what='bf16 32 -> 16 wgrad vs torch'); close(db, b0.grad, rtol=2e-5, what='bf16 32 -> 16 dbias vs torch')
    with hip.option('debug_mode', 3030):
        dw_g, db_g = hip.conv2d_bwd_weight(xd, dyd, 3, 3, 1, 1, need_bias=True)
    close(dw, dw_g, rtol=2e-6, what='vs the generic bf16 kernel'); close(db, db_g, rtol=2e-6, what='dbias vs the generic bf16 kernel')
    assert not torch.equal(dw, dw_g), 'the dedicated kernel did not run'
    sink = torch.full((16,), -1.0, device=dev())
    dw2, none = hip.conv2d_bwd_weight(xd, dyd, 3, 3, 1, 1, need_bias=True, bias_sink=sink)
    assert none is None and torch.equal(dw2, dw) and torch.equal(sink, db - 1.0)


@pytest.mark.parametrize('case', [(2, 203, 171), (4, 128, 160), (1, 256, 256), (5, 129, 130)], ids=str)
def test_data_gradient_16_to_32_six_products(mrdis, case):
    """conv3x3_c16t_split6_kernel (mrdis_c16.hip): the data gradient of sp6.out (dy 16 channels -> dx 32) on the bf16 matrix pipe -- three bf16 terms per fp32
    operand, six products, v_mfma_f32_32x32x16_bf16 (K = the 16 channels); option split6 = 7 (not in the default policy: no gain in the step).  Against torch, against the kernel the layer takes with option split6 = 0 (Winograd
    F(4x4) / direct) and a float64 reference: the six-product form may not be further from float64 than twice the other + 1e-7; ragged tiles both ways."""
    N, H, W = case
    hip = mrdis.hip
    w = rnd((16, 32, 3, 3), 121, 0.15)                     # the 32 -> 16 layer
    dy = rnd((N, 16, H, W), 122)
    want = torch.nn.grad.conv2d_input((N, 32, H, W), w, dy, 1, 1)
    tkc = to_tkc(w).to(dev())
    with hip.option('split6', 0):
        dx0 = hip.conv2d_bwd_data(cl(dy), tkc, (H, W), 3, 3, 1, 1)
    with hip.option('split6', 7):
        dx6 = hip.conv2d_bwd_data(cl(dy), tkc, (H, W), 3, 3, 1, 1)
    assert not torch.equal(dx6, dx0), 'the six-product kernel did not run'          # (split6 = 7 only: the default policy keeps the Winograd kernel for this direction)
    close(dx6, want, rtol=2e-5, what='16 -> 32 six-product dgrad vs torch')
    want64 = torch.nn.grad.conv2d_input((N, 32, H, W), w.double(), dy.double(), 1, 1)
    sc = float(want64.abs().max())
    e0, e6 = float((dx0.cpu().double() - want64).abs().max()) / sc, float((dx6.cpu().double() - want64).abs().max()) / sc
    assert e6 <= 2.0 * e0 + 1e-7, (e6, e0)
    out = hip.empty_nhwc(N, 48, H, W, dev()); out.fill_(3.0)
    with hip.option('split6', 7):
        hip.conv2d_bwd_data(cl(dy), tkc, (H, W), 3, 3, 1, 1, out=out[:, 8:40])
    assert torch.equal(out[:, 8:40], dx6) and bool((out[:, :8] == 3.0).all()) and bool((out[:, 40:] == 3.0).all())


@pytest.mark.parametrize('case', [(3, 32, 150, 256), (5, 64, 100, 128), (9, 128, 64, 64), (2, 32, 256, 256), (30, 128, 63, 64)], ids=str)
def test_wgrad_si_layers_4_to_c(mrdis, case):
    """wgrad_c4_kernel (mrdis_wgrad_s2.hip): weight + bias gradient of the 4 -> 32 / 64 / 128 `si_layers` (3x3 s1 p1) through a
    four-slot ring of input rows, against torch; row counts that do not divide into the workgroups' runs; bias added into a sink;
    option debug_now16 = 1 (the packed-FMA split-K kernel) must agree."""
    N, Co, H, W = case
    hip = mrdis.hip
    x = rnd((N, 4, H, W), 91)
    w = rnd((Co, 4, 3, 3), 92, 0.1).requires_grad_(True); b = rnd((Co,), 93, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b, 1, 1)
    gy = rnd(tuple(y.shape), 94); y.backward(gy)
    dw, db = hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True)
    close(dw, to_tck(w.grad), rtol=3e-4, what='wgrad c4'); close(db, b.grad, rtol=3e-4, what='dbias c4')
    sink = torch.full((Co,), -1.0, device=dev())
    dw2, none = hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True, bias_sink=sink)
    assert none is None and torch.equal(dw2, dw)
    close(sink, b.grad - 1.0, rtol=3e-4, what='dbias sink')
    hip.set_option('debug_now16', 1)
    try:
        dw3, db3 = hip.conv2d_bwd_weight(cl(x), cl(gy), 3, 3, 1, 1, need_bias=True)
    finally:
        hip.set_option('debug_now16', 0)
    close(dw3, dw.cpu(), rtol=3e-4, what='split-K vs c4'); close(db3, db.cpu(), rtol=3e-4, what='split-K vs c4 bias')


def test_spade_bwd_beta_half_in_place(mrdis):
    """instnorm_spade_bwd(fused_gb=True) when the incoming gradient already is channels [C, 2C) of a 2C-channel NHWC buffer (the
    producing data-gradient kernel wrote it there, ops._GroupedConvFn): only dgamma is written, the buffer is returned as [dgamma | dbeta];
    same dz and same buffer contents as the copying path; the data gradient itself lands in the slice (conv2d_bwd_data out = slice)."""
    hip = mrdis.hip
    N, C, H, W = 3, 32, 40, 56
    z = cl(rnd((N, C, H, W), 95)); gamma = cl(rnd((N, C, H, W), 96, 0.3))
    dy = cl(rnd((N, 16, H, W), 97)); w2 = rnd((16, C, 3, 3), 98, 0.1)
    mean = z.mean(dim=(2, 3)).reshape(-1).contiguous(); rstd = (1.0 / (z.var(dim=(2, 3), unbiased=False) + 1e-5).sqrt()).reshape(-1).contiguous()
    buf = hip.empty_nhwc(N, 2 * C, H, W, dev()); buf.fill_(9.0)
    dmix_view = buf[:, C:]
    assert hip.gb_slot(dmix_view) is None               # an untagged buffer: its lower half may belong to another consumer (torch.cat adjoint, skip half)
    buf._mrdis_gb_private = True                        # what ops._GroupedConvFn sets on the buffer it allocates for this
    assert hip.gb_slot(dmix_view) is buf and hip.gb_slot(buf[:, :C]) is None
    hip.conv2d_bwd_data(dy, to_tkc(w2).to(dev()), (H, W), 3, 3, 1, 1, out=dmix_view)
    dmix = hip.conv2d_bwd_data(dy, to_tkc(w2).to(dev()), (H, W), 3, 3, 1, 1)
    assert torch.equal(dmix_view, dmix) and bool((buf[:, :C] == 9.0).all())
    dz_ref, dgb_ref = hip.instnorm_spade_bwd(dmix, z, gamma, mean, rstd, fused_gb=True)
    dz, dgb = hip.instnorm_spade_bwd(dmix_view, z, gamma, mean, rstd, fused_gb=True)
    assert dgb is buf
    assert torch.equal(dz, dz_ref) and torch.equal(dgb, dgb_ref)


@pytest.mark.parametrize('N,C,h,w', [(3, 32, 20, 28), (2, 64, 7, 5), (1, 16, 1, 9), (2, 128, 16, 16)])
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
@pytest.mark.parametrize('slot', [False, True])
def test_spade_bwd_with_the_resize_adjoint_inside(mrdis, N, C, h, w, mode, slot):
    """mrdis_instnorm_spade_bwd_up2: the SPADE backward whose z is a x2 bilinear resize, with the resize's adjoint applied inside the kernel, against the two
    kernels it replaces (mrdis_instnorm_spade_bwd + mrdis_bilinear_bwd): [dgamma | dbeta] bit-identical, d x equal to rounding (fp32: the same expressions in
    the same order; bf16: the two-kernel form rounds the full-resolution d z to bf16 on the way); one-pixel-wide and odd low-resolution maps (every border
    clamp), the in-place beta half."""
    hip = mrdis.hip
    H, W = 2 * h, 2 * w
    T = torch.bfloat16 if mode == 'bf16' else torch.float32
    x = cl(rnd((N, C, h, w), 11)).to(T).contiguous(memory_format=torch.channels_last)
    z = hip.bilinear_fwd(x, (H, W), False)
    gamma = cl(rnd((N, C, H, W), 12, 0.3)).to(T).contiguous(memory_format=torch.channels_last)
    zf = z.float()
    mean = zf.mean(dim=(2, 3)).reshape(-1).contiguous(); rstd = (1.0 / (zf.var(dim=(2, 3), unbiased=False) + 1e-5).sqrt()).reshape(-1).contiguous()
    dsrc = cl(rnd((N, C, H, W), 13)).to(T).contiguous(memory_format=torch.channels_last)

    def grad_in():
        if not slot:
            return dsrc
        buf = hip.empty_nhwc(N, 2 * C, H, W, dev(), T); buf.fill_(9.0); buf._mrdis_gb_private = True
        buf[:, C:].copy_(dsrc)
        return buf[:, C:]
    d1 = grad_in()
    dz, dgb_ref = hip.instnorm_spade_bwd(d1, z, gamma, mean, rstd, fused_gb=True)
    dx_ref = hip.bilinear_bwd(dz, (h, w), False)
    d2 = grad_in()
    res = hip.instnorm_spade_bwd(d2, z, gamma, mean, rstd, fused_gb=True, up2=True)
    assert res is not None
    dx, dgb = res
    assert dx.shape == x.shape and dgb.shape == dgb_ref.shape
    if slot:
        assert dgb is d2._base
    assert torch.equal(dgb, dgb_ref)
    ref64 = None
    if mode == 'f32':
        close(dx, dx_ref, rtol=2e-6, what='dx')
    else:
        ref64 = hip.bilinear_bwd(hip.instnorm_spade_bwd(d1.float(), z.float(), gamma.float(), mean, rstd, fused_gb=True)[0], (h, w), False)
        e_new = float((dx.float() - ref64).abs().max()); e_old = float((dx_ref.float() - ref64).abs().max())
        assert e_new <= 1.5 * e_old + 1e-3 * float(ref64.abs().max()), (e_new, e_old)
    # z not stored: both passes interpolate it from x as the forward kernel formed it
    d3 = grad_in()
    dx3, dgb3 = hip.instnorm_spade_bwd(d3, None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x)
    if mode == 'f32':
        close(dgb3, dgb_ref, rtol=2e-6, what='dgb from x'); close(dx3, dx_ref, rtol=1e-5, what='dx from x (one pass: U^T dzh, sums, 3 x 3 stencil of x)')
        hip.set_option('debug_mode', 2001)          # the two-pass form with z interpolated
        try:
            dx4, dgb4 = hip.instnorm_spade_bwd(grad_in(), None, gamma, mean, rstd, fused_gb=True, up2=True, xlo=x)
        finally:
            hip.set_option('debug_mode', -1)
        close(dgb4, dgb_ref, rtol=2e-6, what='dgb from x, two passes'); close(dx4, dx_ref, rtol=4e-6, what='dx from x, two passes')
    else:
        close(dgb3, dgb_ref, rtol=1e-2, what='dgb from x')
        e3 = float((dx3.float() - ref64).abs().max())
        assert e3 <= 1.5 * float((dx_ref.float() - ref64).abs().max()) + 1e-3 * float(ref64.abs().max())


def test_gb_spade_takes_the_gradient_of_the_resize_input(mrdis):
    """ops.gb_spade on z = ops.bilinear_up2(x): the node receives z detached and returns d x itself (no full-resolution d z, no bilinear backward node);
    the gradients of x, of the anatomy features and of the filters against the unfused graph."""
    ops, hip = mrdis.ops, mrdis.hip
    B, C, h, w = 2, 32, 12, 10
    x0 = cl(rnd((B, C, h, w), 21)); si0 = cl(rnd((B, C, 2 * h, 2 * w), 22))
    wt = rnd((9, C, 2 * C), 23, 0.05).to(dev()); bias = rnd((2 * C,), 24, 0.1).to(dev())
    wk = wt.permute(0, 2, 1).contiguous()
    outs = []
    for fused in (False, True):
        ops.set_up2_bwd_fused(fused)
        x = x0.clone().requires_grad_(True); si = si0.clone().requires_grad_(True)
        a, b = wt.clone().requires_grad_(True), wk.clone()
        z = ops.bilinear_up2(x, 1e-5)
        mix = ops.gb_spade(si, z, [(a, b)], bias, 1e-5)
        (mix * cl(rnd(tuple(mix.shape), 25))).sum().backward()
        outs.append((mix.detach(), x.grad, si.grad, a.grad))
    ops.set_up2_bwd_fused(True)
    assert torch.equal(outs[0][0], outs[1][0])
    close(outs[1][1], outs[0][1], rtol=2e-6, what='dx'); close(outs[1][2], outs[0][2], rtol=2e-6, what='dsi'); close(outs[1][3], outs[0][3], rtol=2e-6, what='dw')


@pytest.mark.parametrize('case', [(32, 256, 256, 16, 16, 4), (8, 64, 128, 64, 64, 4), (6, 128, 128, 16, 16, 3), (5, 32, 64, 37, 29, 3), (3, 16, 32, 128, 128, 3)], ids=str)
@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_stride2_dgrad_parity_classes_in_one_launch(mrdis, case, mode):
    """The data gradient of a stride-2 layer runs as four output-parity classes; they share ONE launch (tapconv_pack_kernel /
    bconv_pack_kernel, blockIdx.y = class).  Against the four-launch form (option debug_nopack = 1): bit-identical (same kernel body,
    same per-class block mapping), odd extents (classes of different sizes) and 3x3 taps (1 / 2 / 2 / 4 taps per class) included;
    fp32 also against torch."""
    N, Ci, Co, H, W, k = case
    hip = mrdis.hip
    w = rnd((Co, Ci, k, k), 101, 0.05)
    Ho, Wo = hip.conv_out_hw(H, W, k, k, 2, 1)
    gy = rnd((N, Co, Ho, Wo), 102)
    w_tkc = to_tkc(w).to(dev())
    if mode == 'bf16':
        dy = cl(gy).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = hip.cast_bf16(to_tck(w).to(dev()))
        kw_ = dict(w_bf16=wb)
    else:
        dy, kw_ = cl(gy), {}
    dx = hip.conv2d_bwd_data(dy, w_tkc, (H, W), k, k, 2, 1, **kw_)
    hip.set_option('debug_nopack', 1)
    try:
        dx4 = hip.conv2d_bwd_data(dy, w_tkc, (H, W), k, k, 2, 1, **kw_)
    finally:
        hip.set_option('debug_nopack', 0)
    assert torch.equal(dx, dx4), 'one launch vs four launches'
    if mode == 'f32':
        want = torch.nn.grad.conv2d_input((N, Ci, H, W), w, gy, 2, 1)
        close(dx, want, rtol=3e-5, what='stride-2 dgrad')


def test_conv_large_grid_256_position_tiles(mrdis):
    """a 32-cout layer with >= 4096 workgroups takes the 256-position tile variant of tapconv_kernel (forward
    and data gradient), ragged in both image dimensions.  wino=0: under the default policy this grid would go to the
    Winograd kernel and the BM=256 direct variant (the fallback on every big layer) would not run at all."""
    hip = mrdis.hip
    hip.set_option('wino', 0)
    N, Ci, Co, H, W = 12, 64, 32, 200, 216
    x = rnd((N, Ci, H, W), 90); w = rnd((Co, Ci, 3, 3), 91, 0.1); b = rnd((Co,), 92, 0.1)
    want = F.conv2d(x, w, b, 1, 1)
    got = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1)
    close(got, want, what='fwd BM256')
    gy = rnd((N, Co, H, W), 93)
    w2 = rnd((Co, Ci, 3, 3), 94, 0.1)                       # dgrad of a Ci=64 <- Co=32 layer is a 32 -> 64 conv: use the transpose
    want_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), w2, gy, 1, 1)
    dx = hip.conv2d_bwd_data(cl(gy), to_tkc(w2).to(dev()), (H, W), 3, 3, 1, 1)
    close(dx, want_dx, what='dgrad')
    x2 = rnd((N, Co, H, W), 95); w3 = rnd((Ci, Co, 3, 3), 96, 0.1)       # 32 <- 64 data gradient: Cout' = 32 -> BM 256 path
    gy2 = rnd((N, Ci, H, W), 97)
    want_dx2 = torch.nn.grad.conv2d_input((N, Co, H, W), w3, gy2, 1, 1)
    dx2 = hip.conv2d_bwd_data(cl(gy2), to_tkc(w3).to(dev()), (H, W), 3, 3, 1, 1)
    close(dx2, want_dx2, what='dgrad BM256')


def test_conv_strided_views(mrdis):
    """channel slices of wider NHWC buffers as input and output (ld != C)."""
    hip = mrdis.hip
    N, H, W = 2, 24, 32
    big = rnd((N, 28, H, W), 5)
    w = rnd((32, 7, 4, 4), 6, 0.2); b = rnd((32,), 7, 0.1)
    bigd = cl(big)
    xs = bigd[:, 7:14]                                   # ld = 28, channel offset 7 (unaligned -> scalar path)
    want = F.conv2d(big[:, 7:14], w, b, 2, 1)
    out_big = torch.zeros((N, 48, H // 2, W // 2), device=dev()).contiguous(memory_format=torch.channels_last)
    hip.conv2d_fwd(xs, to_tck(w).to(dev()), b.to(dev()), 4, 4, 2, 1, out=out_big[:, 16:48])
    close(out_big[:, 16:48], want, what='strided fwd')
    assert float(out_big[:, :16].abs().max()) == 0.0
    gy = rnd(tuple(want.shape), 8)
    gyd = torch.zeros((N, 40, H // 2, W // 2), device=dev()).contiguous(memory_format=torch.channels_last)
    gyd[:, 8:40] = gy.to(dev())
    xr = big[:, 7:14].clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, b, 2, 1).backward(gy)
    dw, db = hip.conv2d_bwd_weight(xs, gyd[:, 8:40], 4, 4, 2, 1)
    close(dw, to_tck(wr.grad), rtol=2e-4, what='strided wgrad')
    dx = hip.conv2d_bwd_data(gyd[:, 8:40], to_tkc(w).to(dev()), (H, W), 4, 4, 2, 1)
    close(dx, xr.grad, what='strided dgrad')


def test_mix_experts(mrdis):
    hip = mrdis.hip
    W = rnd((3, 6, 5, 3, 3), 9).requires_grad_(True)
    r = torch.tensor([0.3, 0.6, 0.9], requires_grad=True)
    mixed = (r[:, None, None, None, None] * W).sum(0)
    g = rnd((9, 5, 6), 10)
    (to_tck(mixed) * g).sum().backward()
    tck, tkc = hip.mix_experts_fwd(W.detach().to(dev()), r.detach().to(dev()))
    close(tck, to_tck(mixed), rtol=1e-6); close(tkc, to_tkc(mixed), rtol=1e-6)
    dW, dr = hip.mix_experts_bwd(g.to(dev()), W.detach().to(dev()), r.detach().to(dev()))
    close(dW, W.grad, rtol=1e-6); close(dr, r.grad, rtol=1e-5)


def test_mix_experts_routed(mrdis):
    """routing sigmoid(Linear(type)) fused into the mix kernels, gradients of the routing Linear included."""
    hip = mrdis.hip
    W = rnd((3, 6, 5, 3, 3), 70).requires_grad_(True)
    fcw = rnd((3, 1), 71).requires_grad_(True); fcb = rnd((3,), 72).requires_grad_(True)
    t = torch.tensor([[3.0]])
    r = torch.sigmoid(F.linear(t, fcw, fcb))[0]
    mixed = (r[:, None, None, None, None] * W).sum(0)
    g = rnd((9, 5, 6), 73)
    (to_tck(mixed) * g).sum().backward()
    tck, tkc, rd = hip.mix_experts_routed_fwd(W.detach().to(dev()), fcw.detach().to(dev()), fcb.detach().to(dev()), t.to(dev()))
    close(rd, r, rtol=1e-6); close(tck, to_tck(mixed), rtol=1e-6); close(tkc, to_tkc(mixed), rtol=1e-6)
    dW, dfcw, dfcb = hip.mix_experts_routed_bwd(g.to(dev()), W.detach().to(dev()), rd, t.to(dev()), 1)
    close(dW, W.grad, rtol=1e-6); close(dfcw, fcw.grad, rtol=1e-5); close(dfcb, fcb.grad, rtol=1e-5)


def test_mix_experts_routed_multi(mrdis):
    """all modality labels of a layer in one launch pair; a label without gradient (None) contributes nothing."""
    hip = mrdis.hip
    W = rnd((3, 40, 24, 3, 3), 80).requires_grad_(True)
    fcw = rnd((3, 1), 81).requires_grad_(True); fcb = rnd((3,), 82).requires_grad_(True)
    types = torch.tensor([[1.0], [2.0], [3.0], [4.0]])
    gs = [rnd((9, 24, 40), 83), None, rnd((9, 24, 40), 84), rnd((9, 24, 40), 85)]
    loss = 0
    mixed_ref = []
    for m in range(4):
        r = torch.sigmoid(F.linear(types[m:m + 1], fcw, fcb))[0]
        mixed = (r[:, None, None, None, None] * W).sum(0)
        mixed_ref.append(mixed)
        if gs[m] is not None:
            loss = loss + (to_tck(mixed) * gs[m]).sum()
    loss.backward()
    tck, tkc, rd = hip.mix_experts_routed_multi_fwd(W.detach().to(dev()), fcw.detach().to(dev()), fcb.detach().to(dev()), types.to(dev()))
    for m in range(4):
        close(tck[m], to_tck(mixed_ref[m]), rtol=1e-6); close(tkc[m], to_tkc(mixed_ref[m]), rtol=1e-6)
    dW, dfcw, dfcb = hip.mix_experts_routed_multi_bwd([None if g is None else g.to(dev()) for g in gs], W.detach().to(dev()), rd, types.to(dev()))
    close(dW, W.grad, rtol=1e-5); close(dfcw, fcw.grad, rtol=1e-5); close(dfcb, fcb.grad, rtol=1e-5)
    # the autograd op: same numbers through torch.autograd, unused labels allowed
    Wd = W.detach().to(dev()).requires_grad_(True); fwd_ = fcw.detach().to(dev()).requires_grad_(True); fbd = fcb.detach().to(dev()).requires_grad_(True)
    outs = mrdis.ops.mix_experts_routed_all(Wd, fwd_, fbd, types.to(dev()))
    sum((outs[2 * m] * gs[m].to(dev())).sum() for m in range(4) if gs[m] is not None).backward()
    close(Wd.grad, W.grad, rtol=1e-5); close(fwd_.grad, fcw.grad, rtol=1e-5); close(fbd.grad, fcb.grad, rtol=1e-5)


@pytest.mark.parametrize('C,N,H,W', [(64, 2, 12, 16), (32, 3, 9, 7), (256, 2, 5, 6), (16, 2, 8, 8), (48, 2, 6, 10)])
def test_batchnorm(mrdis, C, N, H, W):
    hip = mrdis.hip
    x = (rnd((N, C, H, W), 11) * 2 + 0.5).requires_grad_(True)
    gm = rnd((C,), 12).requires_grad_(True); bt = rnd((C,), 13).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.batch_norm(x, rm, rv, gm, bt, True, 0.1, 1e-5)
    gy = rnd(tuple(y.shape), 14); y.backward(gy)
    rmd, rvd = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    yd, mean, rstd = hip.bn_train_fwd(cl(x.detach()), gm.detach().to(dev()), bt.detach().to(dev()), rmd, rvd, 1e-5, 0.1)
    close(yd, y); close(rmd, rm, rtol=1e-5); close(rvd, rv, rtol=1e-5)
    dx, dg, db = hip.bn_train_bwd(cl(gy), cl(x.detach()), gm.detach().to(dev()), mean, rstd)
    close(dx, x.grad, rtol=2e-4); close(dg, gm.grad, rtol=2e-4); close(db, bt.grad, rtol=2e-4)


@pytest.mark.parametrize('C,N,H,W', [(32, 2, 12, 10), (7, 3, 5, 6), (256, 2, 5, 6)])
def test_batchnorm_eval(mrdis, C, N, H, W):
    """running-statistics BatchNorm of the inference path (model.eval())."""
    hip = mrdis.hip
    x = rnd((N, C, H, W), 21) * 2 + 0.5
    gm, bt = rnd((C,), 22), rnd((C,), 23)
    rm, rv = rnd((C,), 24) * 0.3, rnd((C,), 25).abs() + 0.2
    y = F.batch_norm(x, rm, rv, gm, bt, False, 0.1, 1e-5)
    yd = hip.bn_eval_fwd(cl(x), gm.to(dev()), bt.to(dev()), rm.to(dev()), rv.to(dev()), 1e-5)
    close(yd, y)


@pytest.mark.parametrize('C,N,H,W', [(128, 2, 5, 6), (32, 2, 16, 24), (64, 3, 10, 12)])
def test_instnorm_spade(mrdis, C, N, H, W):
    hip = mrdis.hip
    z = (rnd((N, C, H, W), 15) * 1.5 + 0.3).requires_grad_(True)
    g = rnd((N, C, H, W), 16).requires_grad_(True); b = rnd((N, C, H, W), 17).requires_grad_(True)
    out = F.instance_norm(z, eps=1e-5) * (1 + g) + b
    go = rnd(tuple(out.shape), 18); out.backward(go)
    od, mean, rstd = hip.instnorm_spade_fwd(cl(z.detach()), cl(g.detach()), cl(b.detach()), 1e-5)
    close(od, out)
    dz, dg = hip.instnorm_spade_bwd(cl(go), cl(z.detach()), cl(g.detach()), mean, rstd)
    close(dz, z.grad, rtol=2e-4); close(dg, g.grad, rtol=2e-4)


@pytest.mark.parametrize('shape,out_hw,ac', [
    ((2, 8, 5, 6), (10, 12), True), ((2, 8, 5, 6), (10, 12), False), ((2, 4, 32, 64), (5, 6), False),
    ((2, 4, 40, 48), (40, 48), False), ((1, 3, 7, 9), (14, 18), True), ((2, 4, 32, 48), (16, 24), False),
    ((2, 5, 6, 6), (12, 12), False),
    # >= 6 M elements: the branch-free backward kernel (x2 up-sampling both conventions: 5x5 support; down-sampling: 3x3)
    ((4, 32, 224, 224), (448, 448), True), ((4, 32, 224, 216), (448, 432), False), ((2, 32, 320, 328), (160, 164), False)])
def test_bilinear(mrdis, shape, out_hw, ac):
    hip = mrdis.hip
    x = rnd(shape, 19).requires_grad_(True)
    y = F.interpolate(x, size=out_hw, mode='bilinear', align_corners=ac)
    gy = rnd(tuple(y.shape), 20); y.backward(gy)
    close(hip.bilinear_fwd(cl(x.detach()), out_hw, ac), y, rtol=1e-5)
    close(hip.bilinear_bwd(cl(gy), shape[2:], ac), x.grad, rtol=1e-5)


def test_bilinear_golden(mrdis, golden_dir):
    import os
    u = np.load(os.path.join(golden_dir, 'units.npz'))
    hip = mrdis.hip
    x = rnd((2, 3, 5, 6), 13)
    close(hip.bilinear_fwd(cl(x), (10, 12), True), torch.from_numpy(u['bil_ac_true_x2']), rtol=1e-5)
    close(hip.bilinear_fwd(cl(x), (10, 12), False), torch.from_numpy(u['bil_ac_false_x2']), rtol=1e-5)
    xs = rnd((2, 4, 32, 64), 14)
    close(hip.bilinear_fwd(cl(xs), (5, 6), False), torch.from_numpy(u['bil_down_to_5x6']), rtol=1e-5)


def test_softmax_mask_drop(mrdis):
    hip = mrdis.hip
    s = rnd((2, 4, 9, 11), 21, 3.0).requires_grad_(True)
    m = (rnd((2, 9, 11), 22) > 0.5).float()
    out = F.softmax(torch.cat([100 * m.unsqueeze(1), s], 1), 1)[:, 1:]
    go = rnd(tuple(out.shape), 23); out.backward(go)
    od = hip.softmax_mask_drop_fwd(cl(s.detach()), m.to(dev()), 100.0)
    close(od, out, rtol=1e-5)
    close(hip.softmax_mask_drop_bwd(cl(go), od), s.grad, rtol=1e-4)


@pytest.mark.parametrize('p', [1, 2])
def test_recon_err(mrdis, p):
    hip = mrdis.hip
    gt = rnd((3, 7, 12, 10), 24); x = rnd((3, 7, 12, 10), 25).requires_grad_(True)
    d = gt - x
    e = d.abs().mean((1, 2, 3)) if p == 1 else d.pow(2).mean((1, 2, 3))
    w = rnd((3,), 26); (e * w).sum().backward()
    close(hip.recon_err_fwd(cl(gt), cl(x.detach()), p), e, rtol=1e-5)
    close(hip.recon_err_bwd(cl(gt), cl(x.detach()), w.to(dev()), p), x.grad, rtol=1e-5)


def test_maxpool(mrdis):
    hip = mrdis.hip
    x = rnd((2, 4, 32, 48), 27).requires_grad_(True)
    y = F.max_pool2d(x, 16); gy = rnd(tuple(y.shape), 28); y.backward(gy)
    yd, arg = hip.maxpool_fwd(cl(x.detach()), 16)
    close(yd, y, rtol=0, atol=0)
    close(hip.maxpool_bwd(cl(gy), arg, tuple(x.shape), 16), x.grad, rtol=0, atol=0)


def test_lrelu_bwd(mrdis):
    hip = mrdis.hip
    x = rnd((2, 8, 5, 7), 29).requires_grad_(True)
    y = F.leaky_relu(x, 0.2); gy = rnd(tuple(y.shape), 30); y.backward(gy)
    close(hip.lrelu_bwd(cl(gy), cl(y.detach()), 0.2), x.grad, rtol=0, atol=0)


def test_adam_amsgrad_clip(mrdis):
    hip = mrdis.hip
    n = 10007
    p0 = rnd((n,), 31); grads = [rnd((n,), 32 + i, 3.0) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-4, weight_decay=1e-5, amsgrad=True)
    pd = p0.clone().to(dev()); m = torch.zeros(n, device=dev()); v = torch.zeros(n, device=dev()); vm = torch.zeros(n, device=dev())
    nf = torch.zeros(2, device=dev())
    for i, g in enumerate(grads):
        pr.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = g.to(dev()); nf.zero_(); hip.sumsq_finite(gd, nf)
        assert abs(float(nf[0].sqrt()) - float(tn)) <= 1e-5 * float(tn) and float(nf[1]) == 0
        hip.adam_amsgrad_step(pd, gd, m, v, vm, 2e-4, 0.9, 0.999, 1e-8, 1e-5, i + 1, nf, 1.0)
    close(pd, pr, rtol=1e-6, atol=1e-7)
    # a non-finite gradient makes the device skip the step
    gd = grads[0].to(dev()); gd[5] = float('nan'); nf.zero_(); hip.sumsq_finite(gd, nf)
    before = pd.clone()
    hip.adam_amsgrad_step(pd, gd, m, v, vm, 2e-4, 0.9, 0.999, 1e-8, 1e-5, 4, nf, 1.0)
    assert float(nf[1]) == 1 and torch.equal(before, pd)


def test_error_codes(mrdis):
    hip = mrdis.hip
    lib = hip.load()
    assert lib.mrdis_strerror(-2) == b'unsupported geometry'
    x = cl(rnd((1, 4, 8, 8), 40))
    with pytest.raises(hip.MrdisError):
        hip.conv2d_fwd(x, torch.zeros((25, 4, 8), device=dev()), None, 5, 5, 1, 2)    # 25 taps > MRDIS_MAX_TAPS


@pytest.mark.parametrize('N,C,Ct,H,W', [(3, 7, 28, 20, 24), (2, 1, 1, 160, 192), (2, 7, 7, 9, 300)])
def test_recon_metrics_vs_unpinned_host_restatement(mrdis, N, C, Ct, H, W):
    """evaluate() metrics (util.py:935-978) on the device vs the host restatement of the skimage formulas -- which is NOT pinned by
    the reference (scikit-image is absent from this image, see oracle/ref_model.py): device == restatement, nothing more; target is a channel
    slice of a wider NHWC tensor as in EvalStep."""
    from oracle import ref_model as R
    tfull = rnd((N, Ct, H, W), 31); tfull[:, :, :3] = -10.0
    c0 = Ct - C
    t = tfull[:, c0:c0 + C]
    p = t + 0.25 * rnd((N, C, H, W), 32)
    want = R.ref_reconstruction_metrics(t.numpy(), p.numpy())
    got = mrdis.hip.recon_metrics(cl(tfull)[:, c0:c0 + C], cl(p)).cpu().numpy()
    np.testing.assert_allclose(got[:, 0], want['rmse'], rtol=1e-5)
    np.testing.assert_allclose(got[:, 1], want['psnr'], rtol=1e-5)
    np.testing.assert_allclose(got[:, 2], want['ssim'], rtol=1e-5, atol=1e-6)


def test_torch_ops_registered_with_fake_and_autograd(mrdis):
    """torch.ops.mrdis.* (north_star: "exposed ... as custom torch.ops"): schema, CUDA kernel, fake kernel and autograd
    formula pass torch.library.opcheck; `cond_conv2d` -- the reference's CondConv2d.forward seam (model.py:2108-2117) --
    matches the per-sample restatement in plain torch, forward and all five gradients; CondConv2d.forward routes through it."""
    from torch.library import opcheck
    N, Ci, Co, H, W = 2, 8, 16, 12, 10
    x = cl(rnd((N, Ci, H, W), 1)).requires_grad_(True)
    Wt = rnd((3, Co, Ci, 3, 3), 2, 0.2).to(dev()).requires_grad_(True)
    fcw = rnd((3, 1), 3).to(dev()).requires_grad_(True); fcb = rnd((3,), 4).to(dev()).requires_grad_(True)
    b = rnd((Co,), 5, 0.1).to(dev()).requires_grad_(True)
    t = torch.tensor([[2.0]], device=dev())
    tests = ('test_schema', 'test_autograd_registration', 'test_faketensor', 'test_aot_dispatch_dynamic')
    opcheck(torch.ops.mrdis.cond_conv2d.default, (x, t, Wt, fcw, fcb, b, 1, 1, True), test_utils=tests)
    w_tck, w_tkc, r = torch.ops.mrdis.mix_experts_routed(Wt, fcw, fcb, t)
    opcheck(torch.ops.mrdis.mix_experts_routed.default, (Wt, fcw, fcb, t), test_utils=tests)
    opcheck(torch.ops.mrdis.conv2d.default, (x, w_tck.detach().requires_grad_(True), w_tkc.detach(), b, 3, 3, 1, 1, False), test_utils=tests)
    opcheck(torch.ops.mrdis.conv2d.default, (x, w_tck.detach(), w_tkc.detach(), None, 3, 3, 1, 1, True), test_utils=tests)
    gy = cl(rnd((N, Co, H, W), 6))
    opcheck(torch.ops.mrdis.conv2d_bwd_data.default, (gy, w_tkc.detach(), H, W, 3, 3, 1, 1), test_utils=('test_schema', 'test_faketensor'))
    opcheck(torch.ops.mrdis.conv2d_bwd_weight.default, (x.detach(), gy, 3, 3, 1, 1, True), test_utils=('test_schema', 'test_faketensor'))
    sink = torch.zeros(Co, device=dev())
    opcheck(torch.ops.mrdis.conv2d_bwd_weight_sink.default, (x.detach(), gy, 3, 3, 1, 1, sink), test_utils=('test_schema', 'test_faketensor'))
    # numbers: the seam op vs the reference's formula in plain torch on the host
    xc = x.detach().cpu().requires_grad_(True); Wc = Wt.detach().cpu().requires_grad_(True)
    fwc = fcw.detach().cpu().requires_grad_(True); fbc = fcb.detach().cpu().requires_grad_(True); bc = b.detach().cpu().requires_grad_(True)
    rr = torch.sigmoid(F.linear(t.cpu(), fwc, fbc))[0]
    want = F.leaky_relu(F.conv2d(xc, (rr[:, None, None, None, None] * Wc).sum(0), bc, 1, 1), 0.2)
    want.backward(gy.cpu())
    got = torch.ops.mrdis.cond_conv2d(x, t, Wt, fcw, fcb, b, 1, 1, True)
    got.backward(gy)
    close(got, want, what='cond_conv2d'); close(x.grad, xc.grad, what='dx'); close(Wt.grad, Wc.grad, rtol=2e-4, what='dW')
    close(fcw.grad, fwc.grad, rtol=5e-4, what='dfc.w'); close(fcb.grad, fbc.grad, rtol=5e-4, what='dfc.b'); close(b.grad, bc.grad, rtol=2e-4, what='db')
    # the module routes through the op (no step cache active): same result from CondConv2d.forward
    mod = mrdis.CondConv2d(Ci, Co, 3, 1, padding=1).to(dev())
    with torch.no_grad():
        mod.weight.copy_(Wt); mod._routing_fn.fc.weight.copy_(fcw); mod._routing_fn.fc.bias.copy_(fcb); mod.bias.copy_(b)
    calls = []
    orig = torch.ops.mrdis.cond_conv2d

    class Spy:
        def __call__(self, *a, **k):
            calls.append(1); return orig(*a, **k)
    torch.ops.mrdis.cond_conv2d = Spy()
    try:
        y_mod = mod(x.detach(), t.expand(N, 1), lrelu=True)
    finally:
        torch.ops.mrdis.cond_conv2d = orig
    assert calls and torch.equal(y_mod, got.detach())
    # shape propagation without touching the device: fake tensors
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty((4, Ci, 33, 21), device='cuda').contiguous(memory_format=torch.channels_last)
        fy = torch.ops.mrdis.cond_conv2d(fx, torch.empty((1, 1), device='cuda'), torch.empty((3, Co, Ci, 4, 4), device='cuda'),
                                         torch.empty((3, 1), device='cuda'), torch.empty((3,), device='cuda'), None, 2, 1, False)
        assert tuple(fy.shape) == (4, Co, 16, 10) and fy.is_contiguous(memory_format=torch.channels_last)


BF16_CASES = [(2, 32, 64, 40, 56, 3, 1, 1), (3, 64, 32, 33, 21, 3, 1, 1), (20, 128, 256, 16, 16, 3, 1, 1), (4, 16, 48, 20, 24, 3, 1, 1),
              (4, 32, 16, 64, 48, 3, 1, 1), (2, 32, 64, 32, 48, 4, 2, 1), (70, 128, 128, 8, 8, 3, 1, 1), (2, 64, 128, 24, 24, 3, 2, 1),
              (5, 512, 128, 32, 32, 3, 1, 1), (6, 64, 64, 24, 40, 1, 1, 0), (3, 96, 40, 50, 30, 3, 1, 1)]


@pytest.mark.parametrize('case', BF16_CASES, ids=[str(c) for c in BF16_CASES])
def test_conv_bf16_mfma(mrdis, case):
    """MRDIS_DT_F32_BF16M (BASELINE configs[2], stage 1): bf16 MFMA operands, fp32 accumulation, fp32 activations in HBM.
    The kernels must reproduce -- to fp32 rounding -- a torch fp32 convolution of the bf16-ROUNDED operands (that is exactly
    what they multiply), forward / data gradient / weight gradient; against the unrounded fp32 result the error is the bf16
    operand rounding: stated tolerance 1e-2 of the output's max magnitude (measured ~2.5e-3)."""
    N, Ci, Co, H, W, k, s, p = case
    hip = mrdis.hip
    x = rnd((N, Ci, H, W), 1); w = rnd((Co, Ci, k, k), 2, 1.0 / np.sqrt(Ci * k * k)); b = rnd((Co,), 3, 0.1)
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    want = F.conv2d(xb, wb, b, s, p)
    got = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), k, k, s, p, w_bf16=hip.cast_bf16(to_tkc(w).to(dev())))
    close(got, want, rtol=2e-5, what='fwd vs bf16-rounded operands')
    close(got, F.conv2d(x, w, b, s, p), rtol=1e-2, what='fwd vs fp32')
    exact = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), k, k, s, p)
    assert not torch.equal(exact, got)                                            # the bf16 kernel really ran
    gy = rnd(tuple(want.shape), 4); gyb = gy.bfloat16().float()
    dx = hip.conv2d_bwd_data(cl(gy), to_tkc(w).to(dev()), (H, W), k, k, s, p, w_bf16=hip.cast_bf16(to_tck(w).to(dev())))
    if Co % 16 == 0 and Ci % 4 == 0 and Ci >= 16:                                 # the data gradient reduces over Co
        close(dx, torch.nn.grad.conv2d_input(x.shape, wb, gyb, s, p), rtol=2e-5, what='dgrad vs bf16-rounded operands')
    else:                                                                         # not a multiple of 16: the fp32 kernel ran, exact
        close(dx, torch.nn.grad.conv2d_input(x.shape, w, gy, s, p), rtol=1e-4, what='dgrad (fp32 kernel)')
    wz = torch.zeros(Co, Ci, k, k, requires_grad=True)
    F.conv2d(xb, wz, None, s, p).backward(gyb)
    dw, db = hip.conv2d_bwd_weight(cl(x), cl(gy), k, k, s, p, need_bias=True, dtype=hip.DT_F32_BF16M)
    want_b = to_tck(wz.grad)
    wz.grad = None; F.conv2d(x, wz, None, s, p).backward(gy)
    want_f = to_tck(wz.grad)
    err_b = float((dw.cpu() - want_b).abs().max() / want_b.abs().max()); err_f = float((dw.cpu() - want_f).abs().max() / want_f.abs().max())
    if s == 1 and N * H * W >= 4096 and Ci % 32 == 0 and (Co > 32 or Ci >= 128):  # the bf16 kernel's domain (stride-1 "same" layers on real maps)
        assert err_b <= 4e-5, ('wgrad vs bf16-rounded operands', err_b)
    else:                                                                         # stride 2 / tiny maps: the fp32 kernels run -> exact fp32
        assert err_f <= 3e-4, ('wgrad (fp32 kernels)', err_f, err_b)
    close(db, gy.sum((0, 2, 3)), rtol=2e-5, what='dbias (fp32 sum)')


@pytest.mark.parametrize('case', [(8, 64, 128, 64, 64), (5, 32, 64, 50, 70), (3, 128, 256, 33, 47), (16, 64, 32, 32, 32), (2, 32, 40, 96, 128), (40, 128, 128, 16, 16)], ids=str)
def test_bf16_weight_gradient_lds_dma_bit_identical(mrdis, case):
    """bwgrad3_kernel (mrdis_bf16.hip, round 5: both images by LDS-DMA into a ring of three stages, dbias from the LDS image) against bwgrad2_kernel
    (register staging; option debug_mode 3010 keeps it): dw and dbias BIT-IDENTICAL on bf16 views -- ragged tiles, a cout tail (40), every (ci, co) block shape,
    many-image tiles; and against torch fp32 on the bf16-valued operands up to the order of the fp32 sums."""
    N, Ci, Co, H, W = case
    hip = mrdis.hip
    B16 = torch.bfloat16
    x = rnd((N, Ci, H, W), 1).to(B16); gy = rnd((N, Co, H, W), 2).to(B16)
    xd, gyd = cl(x), cl(gy)
    with hip.option('debug_mode', 3010):
        dw2, db2 = hip.conv2d_bwd_weight(xd, gyd, 3, 3, 1, 1, need_bias=True)
    dw3, db3 = hip.conv2d_bwd_weight(xd, gyd, 3, 3, 1, 1, need_bias=True)
    assert torch.equal(dw2, dw3) and torch.equal(db2, db3)
    wz = torch.zeros(Co, Ci, 3, 3, requires_grad=True)
    F.conv2d(x.float(), wz, None, 1, 1).backward(gy.float())
    close(dw3, to_tck(wz.grad), rtol=4e-5, what='bf16 wgrad vs torch fp32 on the bf16-valued operands')
    close(db3, gy.float().sum((0, 2, 3)), rtol=2e-5, what='dbias')


@pytest.mark.parametrize('case', [
    # N, Ci, Co, H, W, k
    (8, 32, 64, 64, 64, 4),          # the anatomy encoder's shape family (4x4 / stride 2 / pad 1)
    (6, 64, 40, 52, 76, 4),          # ragged tiles (26 x 38 class maps), cout tail
    (16, 16, 32, 32, 48, 3),         # Ci = 16: half-filled channel image (modality encoder, 3x3 / stride 2)
    (8, 96, 128, 36, 60, 3),         # three 32-channel blocks, ragged
    (64, 128, 24, 16, 16, 4),        # many-image tiles (8 x 8 class maps)
], ids=lambda c: 'N%d_%dto%d_%dx%d_k%d' % c)
def test_bf16_stride2_weight_gradient_parity_classes(mrdis, case):
    """bwgrad_pack_kernel (mrdis_bf16.hip): the weight gradient of the stride-2 encoder layers on bf16 views, as four stride-1 "same"
    correlations of the input's parity classes in one launch.  bf16 x bf16 products are exact in fp32, so the result equals torch fp32
    on the same (bf16-valued) operands up to the order of the fp32 sums, and the fp32 parity-class kernels between view casts
    (option debug_now16 = 1: the route before this kernel) likewise; the bias gradient is added to a sink when asked."""
    hip = mrdis.hip
    N, Ci, Co, H, W, k = case
    B16 = torch.bfloat16
    x = rnd((N, Ci, H, W), 5).bfloat16(); w0 = torch.zeros(Co, Ci, k, k, requires_grad=True)
    y = F.conv2d(x.float(), w0, None, 2, 1)
    gy = rnd(tuple(y.shape), 6).bfloat16()
    y.backward(gy.float())
    xd, dyd = cl(x.float()).to(B16), cl(gy.float()).to(B16)
    n0 = hip.fallbacks['conv2d_bwd_weight_bf16']
    dw, db = hip.conv2d_bwd_weight(xd, dyd, k, k, 2, 1, need_bias=True)
    assert hip.fallbacks['conv2d_bwd_weight_bf16'] == n0, 'the shape must be inside the bf16 kernel\'s domain'
    close(dw, to_tck(w0.grad), rtol=2e-5, what='stride-2 bf16 wgrad vs torch')
    close(db, gy.float().sum((0, 2, 3)), rtol=2e-5, what='stride-2 bf16 dbias vs torch')
    with hip.option('debug_now16', 1):
        dw1, db1 = hip.conv2d_bwd_weight(xd, dyd, k, k, 2, 1, need_bias=True)
    assert hip.fallbacks['conv2d_bwd_weight_bf16'] == n0 + 1
    close(dw, dw1, rtol=2e-5, what='parity-class bf16 kernel vs fp32 kernels between casts')
    close(db, db1, rtol=2e-5, what='dbias, both routes')
    sink = torch.full((Co,), 2.0, device=dev())
    dw2, none = hip.conv2d_bwd_weight(xd, dyd, k, k, 2, 1, need_bias=True, bias_sink=sink)
    assert none is None and torch.equal(dw2, dw) and torch.equal(sink, db + 2.0)
    # a channel slice of a wider tensor (pixel stride != channels), as the skip concatenation hands over
    wide = torch.cat([xd, xd.flip(1)], 1).contiguous(memory_format=torch.channels_last)
    dw3, _ = hip.conv2d_bwd_weight(wide[:, :Ci], dyd, k, k, 2, 1, need_bias=False)
    assert torch.equal(dw3, dw)


@pytest.mark.parametrize('case', [(2, 32, 256, 256), (9, 64, 128, 128), (32, 128, 64, 64), (3, 32, 250, 256)], ids=lambda c: 'N%d_4to%d_%dx%d' % c)
def test_si_layer_weight_gradient_fp32_map_bf16_gradient(mrdis, case):
    """MRDIS_DT_XF32_YBF16 weight gradient (wgrad_c4_kernel<NT, true>): the SPADE si_layers under bf16 storage multiply the fp32 anatomy
    map with the bf16 gradient of their output on the bf16 matrix pipe (wgrad_c4b_kernel): the map goes in as three bf16 terms (exact), so the
    products are those of the fp32 call on dy.float() and only the order of the fp32 sums differs; option debug_mode 3011 keeps the older form,
    which widens dy and IS the fp32 kernel (bit-identical).  With and without a bias sink; shapes outside the kernel decline (may_decline)."""
    hip = mrdis.hip
    N, Co, H, W = case
    x = cl(rnd((N, 4, H, W), 3))
    dy = cl(rnd((N, Co, H, W), 4)).to(torch.bfloat16)
    got = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, may_decline=True)
    assert got is not None
    want = hip.conv2d_bwd_weight(x, dy.float(), 3, 3, 1, 1, need_bias=True)
    assert got[0].shape == (9, 4, Co)
    close(got[0], want[0], rtol=2e-6, what='bf16-pipe si wgrad vs the fp32 kernel'); close(got[1], want[1], rtol=2e-6, what='bf16-pipe si bias gradient vs the fp32 kernel')
    with hip.option('debug_mode', 3011):
        old = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, may_decline=True)
    assert torch.equal(old[0], want[0]) and torch.equal(old[1], want[1])
    w0 = torch.zeros(Co, 4, 3, 3, requires_grad=True)
    F.conv2d(x.cpu().contiguous(), w0, None, 1, 1).backward(dy.float().cpu().contiguous())
    close(got[0], to_tck(w0.grad), rtol=2e-5, what='si wgrad vs torch')
    sink = torch.full((Co,), -1.0, device=dev())
    dw2, none = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, bias_sink=sink, may_decline=True)
    assert none is None and torch.equal(dw2, got[0]) and torch.equal(sink, got[1] - 1.0)
    dw16, db16 = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, may_decline=True, pad16=True)      # the stored 16-row shape of the filter
    assert dw16.shape == (9, 16, Co) and torch.equal(dw16, F.pad(got[0], (0, 0, 0, 12))) and torch.equal(db16, got[1])
    assert hip.conv2d_bwd_weight(x[:, :, :, :40].contiguous(memory_format=torch.channels_last), dy[:, :, :, :40].contiguous(memory_format=torch.channels_last),
                                 3, 3, 1, 1, may_decline=True) is None


@pytest.mark.parametrize('case', [(2, 64, 256, 256), (8, 32, 128, 128), (32, 64, 64, 64), (3, 32, 250, 256), (5, 64, 101, 256)], ids=lambda c: 'N%d_%dto4_%dx%d' % c)
def test_c_to_4_weight_gradient_bf16_x_fp32_dy(mrdis, case):
    """MRDIS_DT_XBF16_YF32 weight gradient of a C -> 4 3x3 layer (ana_dec.output under bf16 storage; wgrad_c4b_kernel<.., SWAP>): the bf16 trunk times the
    fp32 gradient of the four anatomy logits on the bf16 matrix pipe -- the gradient goes in as three bf16 terms (exact), so against torch fp32 on the same
    operands only the order of the sums differs.  (9, C, 4) in the layer's own tap order, and the four bias sums; with a bias sink; narrow maps decline."""
    hip = mrdis.hip
    N, C, H, W = case
    x = cl(rnd((N, C, H, W), 41)).to(torch.bfloat16)
    dy = cl(rnd((N, 4, H, W), 42))
    got = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, may_decline=True)
    assert got is not None and got[0].shape == (9, C, 4) and got[1].shape == (4,)
    w0 = torch.zeros(4, C, 3, 3, requires_grad=True)
    F.conv2d(x.float().cpu().contiguous(), w0, None, 1, 1).backward(dy.cpu().contiguous())
    close(got[0], to_tck(w0.grad), rtol=2e-5, what='C -> 4 wgrad vs torch')
    close(got[1], dy.cpu().sum((0, 2, 3)), rtol=2e-5, what='C -> 4 bias gradient vs torch')
    sink = torch.full((4,), -1.0, device=dev())
    dw2, none = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, bias_sink=sink, may_decline=True)
    assert none is None and torch.equal(dw2, got[0]) and torch.equal(sink, got[1] - 1.0)
    dw16, db16 = hip.conv2d_bwd_weight(x, dy, 3, 3, 1, 1, need_bias=True, may_decline=True, pad16=True)      # the stored (column-padded) shape of the filter
    assert dw16.shape == (9, C, 16) and torch.equal(dw16, F.pad(got[0], (0, 12))) and torch.equal(db16, got[1])
    assert hip.conv2d_bwd_weight(x[:, :, :, :40].contiguous(memory_format=torch.channels_last), dy[:, :, :, :40].contiguous(memory_format=torch.channels_last),
                                 3, 3, 1, 1, may_decline=True) is None


@pytest.mark.parametrize('case', [(2, 64, 256, 256), (3, 64, 70, 96), (5, 32, 32, 40)], ids=lambda c: 'N%d_%dfrom4_%dx%d' % c)
def test_c_from_4_data_gradient_fp32_dy_bf16_out(mrdis, case):
    """MRDIS_DT_XBF16_YF32 data gradient of a C -> 4 3x3 layer (ana_dec.output under bf16 storage): the Cin = 4 kernel convolves the fp32
    gradient with the reversed taps and writes bf16.  The filter arrives in the 16-row layout [9][16][C] of the padded bf16 kernels.
    The bf16-output form multiplies on the bf16 matrix pipe with both fp32 operands carried as two bf16 terms (2^-16 relative against the fp32
    product) and rounds once at the store: its result is the fp32 call's cast to bf16 except where that sits within 2^-16 of a rounding
    boundary -- well under 1 % of the elements, and then the neighbouring bf16 value."""
    hip = mrdis.hip
    N, C, H, W = case
    w = rnd((4, C, 3, 3), 8, 0.1)
    dy = cl(rnd((N, 4, H, W), 9))
    tkc = to_tkc(w).to(dev())                                       # [9][4][C]
    want = hip.conv2d_bwd_data(dy, tkc, (H, W), 3, 3, 1, 1)
    close(want, torch.nn.grad.conv2d_input((N, C, H, W), w, dy.cpu().contiguous(), 1, 1), rtol=2e-5, what='fp32 C <- 4 dgrad vs torch')
    tkc16 = F.pad(tkc, (0, 0, 0, 12))                               # rows >= 4 zero
    out = hip.empty_nhwc(N, C, H, W, dev(), torch.bfloat16)
    got = hip.conv2d_bwd_data(dy, tkc16, (H, W), 3, 3, 1, 1, out=out, may_decline=True)
    assert got is not None and got.dtype == torch.bfloat16
    ref = want.to(torch.bfloat16)
    differs = (got != ref)
    assert float(differs.float().mean()) < 0.01, float(differs.float().mean())
    gf, rf = got.float(), ref.float()
    # one bf16 step, or -- where the nine taps cancel -- 2^-15 of the scale of what was summed
    assert bool(((gf - rf).abs() <= 2.0 ** -7 * rf.abs() + 2.0 ** -15 * float(want.abs().max())).all()), 'more than one bf16 step away from the fp32 result'
    close(got, want, rtol=4e-3, what='bf16-pipe C <- 4 dgrad vs the fp32 kernel')


def test_conv_bf16_storage_random_shapes(mrdis):
    """MRDIS_DT_BF16 (bf16 activation views in and out) over a seeded sweep of geometries inside the bf16 kernels' domain
    (channels in multiples of 16, >= 16 outputs): no shape may be refused without a working fallback, and every result must
    match torch fp32 on the bf16-rounded operands up to the rounding of the bf16 OUTPUT (2^-8 relative per element)."""
    hip = mrdis.hip
    g = np.random.RandomState(77)
    B16 = torch.bfloat16
    for case in range(30):
        k = int(g.choice([1, 3, 3, 4])); stride = 1 if k == 1 else int(g.choice([1, 1, 2])); pad = 0 if k == 1 else 1
        N = int(g.randint(1, 9)); Ci = 16 * int(g.randint(1, 9)); Co = int(g.choice([16, 20, 32, 48, 64, 96, 128, 256]))
        H = int(g.randint(2, 40)); W = int(g.randint(2, 50))
        if (H + 2 * pad - k) // stride + 1 <= 0 or (W + 2 * pad - k) // stride + 1 <= 0:
            continue
        x = rnd((N, Ci, H, W), 10 + case); w = rnd((Co, Ci, k, k), 100 + case, 1.0 / np.sqrt(Ci * k * k)); b = rnd((Co,), 200 + case, 0.1)
        xb, wb = x.bfloat16().float(), w.bfloat16().float()
        want = F.conv2d(xb, wb, b, stride, pad)
        tag = f'case {case}: N{N} {Ci}->{Co} {H}x{W} k{k} s{stride}'
        xd = cl(x).to(B16)
        got = hip.conv2d_fwd(xd, to_tck(w).to(dev()), b.to(dev()), k, k, stride, pad, w_bf16=hip.cast_bf16(to_tkc(w).to(dev())))
        assert got.dtype == B16
        close(got, want, rtol=6e-3, what=tag + ' fwd')
        gy = rnd(tuple(want.shape), 300 + case); gyb = gy.bfloat16().float()
        dyd = cl(gy).to(B16)
        dx = hip.conv2d_bwd_data(dyd, to_tkc(w).to(dev()), (H, W), k, k, stride, pad, w_bf16=hip.cast_bf16(to_tck(w).to(dev())))
        assert dx.dtype == B16
        close(dx, torch.nn.grad.conv2d_input(x.shape, wb, gyb, stride, pad), rtol=6e-3, what=tag + ' dgrad')
        wz = torch.zeros(Co, Ci, k, k, requires_grad=True)
        F.conv2d(xb, wz, None, stride, pad).backward(gyb)
        dw, db = hip.conv2d_bwd_weight(xd, dyd, k, k, stride, pad, need_bias=True)
        assert dw.dtype == torch.float32
        close(dw, to_tck(wz.grad), rtol=3e-4, what=tag + ' wgrad')
        close(db, gyb.sum((0, 2, 3)), rtol=3e-4, what=tag + ' dbias')


@pytest.mark.parametrize('C,N,H,W', [(64, 2, 12, 16), (32, 3, 9, 7), (256, 2, 5, 6)])
def test_norms_resize_bf16_storage(mrdis, C, N, H, W):
    """BatchNorm (train / eval), InstanceNorm+SPADE, bilinear, LeakyReLU backward on bf16 views: fp32 arithmetic inside,
    so results equal the fp32 formulas on the bf16-rounded inputs up to the rounding of the bf16 outputs; statistics fp32."""
    hip = mrdis.hip
    B16 = torch.bfloat16
    rb = lambda t: t.bfloat16().float()
    x = rb(rnd((N, C, H, W), 11) * 2 + 0.5).requires_grad_(True)
    gm = rnd((C,), 12).requires_grad_(True); bt = rnd((C,), 13).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.batch_norm(x, rm, rv, gm, bt, True, 0.1, 1e-5)
    gy = rb(rnd(tuple(y.shape), 14)); y.backward(gy)
    rmd, rvd = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    xd = cl(x.detach()).to(B16)
    yd, mean, rstd = hip.bn_train_fwd(xd, gm.detach().to(dev()), bt.detach().to(dev()), rmd, rvd, 1e-5, 0.1)
    assert yd.dtype == B16 and mean.dtype == torch.float32
    close(yd, y, rtol=6e-3); close(rmd, rm, rtol=1e-5); close(rvd, rv, rtol=1e-5)
    dx, dg, db = hip.bn_train_bwd(cl(gy).to(B16), xd, gm.detach().to(dev()), mean, rstd)
    close(dx, x.grad, rtol=8e-3); close(dg, gm.grad, rtol=3e-4); close(db, bt.grad, rtol=3e-4)
    close(hip.bn_eval_fwd(xd, gm.detach().to(dev()), bt.detach().to(dev()), rmd, rvd, 1e-5),
          F.batch_norm(x.detach(), rm, rv, gm.detach(), bt.detach(), False, 0.1, 1e-5), rtol=6e-3)
    z = rb(rnd((N, C, H, W), 15) * 1.5 + 0.3).requires_grad_(True)
    g = rb(rnd((N, C, H, W), 16)).requires_grad_(True); b = rb(rnd((N, C, H, W), 17)).requires_grad_(True)
    out = F.instance_norm(z, eps=1e-5) * (1 + g) + b
    go = rb(rnd(tuple(out.shape), 18)); out.backward(go)
    od, m2, r2 = hip.instnorm_spade_fwd(cl(z.detach()).to(B16), cl(g.detach()).to(B16), cl(b.detach()).to(B16), 1e-5)
    close(od, out, rtol=6e-3)
    dz, dgm = hip.instnorm_spade_bwd(cl(go).to(B16), cl(z.detach()).to(B16), cl(g.detach()).to(B16), m2, r2)
    close(dz, z.grad, rtol=8e-3); close(dgm, g.grad, rtol=6e-3)
    xi = rb(rnd((N, C, H, W), 19)).requires_grad_(True)
    for ac in (True, False):
        yi = F.interpolate(xi, size=(2 * H, 2 * W), mode='bilinear', align_corners=ac)
        gyi = rb(rnd(tuple(yi.shape), 20)); xi.grad = None; yi.backward(gyi)
        close(hip.bilinear_fwd(cl(xi.detach()).to(B16), (2 * H, 2 * W), ac), yi, rtol=6e-3)
        close(hip.bilinear_bwd(cl(gyi).to(B16), (H, W), ac), xi.grad, rtol=6e-3)
    ya = F.leaky_relu(xi.detach(), 0.2)
    close(hip.lrelu_bwd(cl(gy).to(B16), cl(rb(ya)).to(B16), 0.2), torch.where(ya > 0, gy, 0.2 * gy), rtol=6e-3)
    t32 = cl(rnd((N, C, H, W), 21))
    assert torch.equal(hip.cast_view(hip.cast_view(t32, B16), torch.float32), t32.bfloat16().float())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('N,C,H,W', [(3, 32, 5, 6), (2, 128, 9, 7), (4, 64, 16, 24), (2, 16, 33, 17)])
def test_bilinear_up2_with_instance_statistics(mrdis, N, C, H, W, dtype):
    """mrdis_bilinear_up2_stats_fwd (nn.Upsample(scale_factor=2, bilinear) in front of a SPADE block, model.py:2551-2573 + the
    InstanceNorm statistics of :2440 in one pass): the map is mrdis_bilinear_fwd's (to the last bit or one: two instantiations of one expression), mean / rstd are those of the STORED
    values (fp64 reference), equal to what mrdis_instnorm_stats computes from the map up to the order of its fp32 partial sums; and
    mrdis_instnorm_spade_fwd with workspace = NULL applies exactly these statistics."""
    hip = mrdis.hip
    T = torch.bfloat16 if dtype == 'bf16' else torch.float32
    x = cl(rnd((N, C, H, W), 41) * 2.0 + 0.7).to(T)
    res = hip.bilinear_up2_stats(x, 1e-5)
    assert res is not None
    y, mean, rstd = res
    y0 = hip.bilinear_fwd(x, (2 * H, 2 * W), False)
    # the same expression compiled in two template instantiations: hipcc may contract its multiply-adds differently (1 ulp)
    assert y.dtype == T and float((y.float() - y0.float()).abs().max()) <= (8e-3 if dtype == 'bf16' else 3e-7) * float(y0.float().abs().max())
    yd = y.double()
    m64 = yd.mean((2, 3)).reshape(-1); v64 = yd.var((2, 3), unbiased=False).reshape(-1)
    assert float((mean.double() - m64).abs().max()) <= 2e-6 * float(m64.abs().max() + 1)
    assert float((rstd.double() - 1 / torch.sqrt(v64 + 1e-5)).abs().max()) <= 2e-5 * float((1 / torch.sqrt(v64 + 1e-5)).max())
    g = cl(rnd((N, C, 2 * H, 2 * W), 42, 0.3)).to(T); b = cl(rnd((N, C, 2 * H, 2 * W), 43, 0.3)).to(T)
    out_a, mean_a, rstd_a = hip.instnorm_spade_fwd(y, g, b, 1e-5)                          # statistics pass of its own
    out_b, mean_b, rstd_b = hip.instnorm_spade_fwd(y, g, b, 1e-5, stats=(mean, rstd))       # the ones that came with the map
    assert mean_b.data_ptr() == mean.data_ptr()
    assert float((mean_a - mean).abs().max()) <= 2e-6 * float(mean.abs().max() + 1) and float((rstd_a - rstd).abs().max()) <= 2e-5 * float(rstd.max())
    tol = 2e-2 if dtype == 'bf16' else 2e-5
    assert float((out_a.float() - out_b.float()).abs().max()) <= tol * float(out_a.float().abs().max())


@pytest.mark.parametrize('R,S,flip,C', [(32, 64, 0, 0), (20, 100, 1, 0), (128, 256, 0, 128), (36, 96, 0, 48), (64, 40, 1, 0)])
def test_winograd_filter_image(mrdis, R, S, flip, C):
    """mrdis_wino_u_jobs: U = G g G^T of a [9][R][S] filter in the order wino2_kernel<.., UIMG> reads it -- [cout tile][chunk of 8][8][4][64][4],
    zero-padded; roles forward (flip 0), data gradient (flip 1: taps reversed) and SPADE (the fused gamma | beta cout order) -- against the
    transform written out in torch."""
    hip = mrdis.hip
    hip.set_option('wino4', 0)                      # the 16-point format for every filter (the 36-point one: tests/test_gpu_wino4.py)
    w = rnd((9, R, S), 77)
    n = hip.wino_u_image_floats(R, S, C)
    img = torch.full((n,), float('nan'), device=dev())
    j = hip.WinoUJob()
    wd = w.to(dev())
    j.w, j.img, j.R, j.S, j.flip, j.spadeC, j.block0, j.nblk = wd.data_ptr(), img.data_ptr(), R, S, flip, C, 0, hip.wino_u_job_blocks(R, S, C)
    hip.wino_u_jobs(hip.wino_u_table([j], dev()), 1, j.nblk)
    G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]], dtype=torch.float64)
    g = w.double().reshape(3, 3, R, S)
    if flip:
        g = g.flip(0, 1)
    U = torch.einsum('ai,ijrs,bj->abrs', G, g, G).reshape(16, R, S)
    tiles = (C + 31) // 32 if C else (S + 63) // 64
    nch = (R + 7) // 8
    want = torch.zeros(tiles, nch, 8, 4, 64, 4, dtype=torch.float64)
    for cot in range(tiles):
        for slot in range(64):
            if C:
                ch = cot * 32 + 16 * (slot >> 5) + (slot & 15)
                co, ok = (C if (slot >> 4) & 1 else 0) + ch, ch < C
            else:
                co = cot * 64 + slot; ok = co < S
            if not ok:
                continue
            col = torch.zeros(16, nch * 8, dtype=torch.float64); col[:, :R] = U[:, :, co]
            want[cot, :, :, :, slot, :] = col.reshape(4, 4, nch, 8).permute(2, 3, 0, 1)          # [chunk][k][a][b]
    got = img.cpu().double().reshape(want.shape)
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('G,B,C,H,W', [(4, 2, 64, 9, 11), (3, 3, 32, 16, 16), (2, 1, 256, 5, 6)])
def test_batchnorm_groups_equal_separate_calls(mrdis, G, B, C, H, W, dtype):
    """mrdis_bn_train_fwd / _bwd with groups = G (the G per-modality calls of one BatchNorm layer, model.py:3135-3157, as one launch
    per kernel): output, saved statistics, running statistics (updated block by block, rounded to fp32 in between), data gradient and
    the parameter-gradient sums are BIT-IDENTICAL to G calls on the sample blocks."""
    hip = mrdis.hip
    T = torch.bfloat16 if dtype == 'bf16' else torch.float32
    x = cl(rnd((G * B, C, H, W), 51) * 1.7 + 0.3).to(T); dy = cl(rnd((G * B, C, H, W), 52)).to(T)
    gamma = (rnd((C,), 53, 0.5) + 1).to(dev()); beta = rnd((C,), 54, 0.2).to(dev())
    rm0 = rnd((C,), 55, 0.1).to(dev()); rv0 = (rnd((C,), 56, 0.1).abs() + 0.5).to(dev())
    rm_a, rv_a = rm0.clone(), rv0.clone()
    ys, dxs, dgs, dbs, means = [], [], [], [], []
    for g in range(G):
        sl = slice(g * B, (g + 1) * B)
        y, mean, rstd = hip.bn_train_fwd(x[sl], gamma, beta, rm_a, rv_a, 1e-5, 0.1)
        dx, dg, db = hip.bn_train_bwd(dy[sl], x[sl], gamma, mean, rstd)
        ys.append(y); dxs.append(dx); dgs.append(dg); dbs.append(db); means.append(torch.cat([mean, rstd]))
    rm_b, rv_b = rm0.clone(), rv0.clone()
    y, mean, rstd = hip.bn_train_fwd(x, gamma, beta, rm_b, rv_b, 1e-5, 0.1, groups=G)
    assert torch.equal(y, torch.cat(ys, 0)) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    assert torch.equal(mean.view(G, C), torch.stack([m_[:C] for m_ in means])) and torch.equal(rstd.view(G, C), torch.stack([m_[C:] for m_ in means]))
    sink_g, sink_b = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    dx, _, _ = hip.bn_train_bwd(dy, x, gamma, mean, rstd, sink=(sink_g, sink_b), groups=G)
    assert torch.equal(dx, torch.cat(dxs, 0))
    acc_g, acc_b = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    for g in range(G):                                  # the order in which separate calls would fold their sums into the sink
        acc_g += dgs[g]; acc_b += dbs[g]
    assert torch.equal(sink_g, acc_g) and torch.equal(sink_b, acc_b)



@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_bilinear_up2_block_scattered_output(mrdis, dtype):
    """mrdis_bilinear_up2_stats_fwd with out_block: image n of the result goes to block n // Bb of a strided block buffer (the shared SPADE
    decoder writes label j's result into column j of the [decoder i][label j] buffer).  Values and statistics are those of the dense call."""
    hip = mrdis.hip
    T = torch.bfloat16 if dtype == 'bf16' else torch.float32
    M, Bb, C, H, W = 4, 3, 32, 10, 12
    x = cl(rnd((M * Bb, C, H, W), 2)).to(T)
    y0, m0, r0 = hip.bilinear_up2_stats(x, 1e-5)
    zbuf = torch.full((M, M * Bb, 2 * H, 2 * W, C), -7.0, dtype=T, device=dev()).permute(0, 1, 4, 2, 3)
    for j in (2, 0):
        y, m1, r1 = hip.bilinear_up2_stats(x, 1e-5, out_blocks=zbuf[:, j * Bb:(j + 1) * Bb])
        assert torch.equal(m1, m0) and torch.equal(r1, r0)
        for i in range(M):
            assert torch.equal(zbuf[i, j * Bb:(j + 1) * Bb], y0[i * Bb:(i + 1) * Bb])
    assert float(zbuf[:, Bb:2 * Bb].float().max()) == -7.0 and float(zbuf[:, 3 * Bb:].float().min()) == -7.0        # other columns untouched


@pytest.mark.parametrize('case', [
    # N, Ci, Co, H, W, k, stride
    (32, 128, 256, 8, 8, 3, 1),        # the 8x8 SPADE level at the bench batch: 32-channel chunks + split-K
    (3, 72, 40, 9, 11, 3, 1),          # ragged half tile (99 positions in 64-position tiles), partial last chunk (72 = 4.5 x 16), cout tail
    (5, 64, 36, 7, 5, 3, 1),           # 35 positions per image: several images per tile
    (32, 256, 256, 16, 16, 4, 2),      # 4x4 stride 2 onto 8x8: 16 taps, 16-channel chunks + split-K
    (2, 96, 64, 12, 12, 1, 1),         # 1x1: the filter slab is smaller than the partial-sum buffer -> no split-K
], ids=lambda c: 'N%d_%dto%d_%dx%d_k%d_s%d' % c)
def test_small_map_direct_conv_split_k(mrdis, case):
    """tapconv_body<.., SK = 2> (half-filled 64-position tiles of the small maps: the two otherwise idle position waves take half of every
    chunk's channels, pairs add through LDS in a fixed order) and the 32-channel chunks of those launches: forward and data gradient
    against torch, and against the same launch without the small-map forms (option debug_now16 = 1: full tiles, one wave per block)
    to fp32 rounding (the split changes the order of the channel sum)."""
    hip = mrdis.hip
    N, Ci, Co, H, W, k, s = case
    p = 0 if k == 1 else 1
    x = rnd((N, Ci, H, W), 21); w = rnd((Co, Ci, k, k), 22, 1.0 / np.sqrt(Ci * k * k)); b = rnd((Co,), 23, 0.1)
    want = F.conv2d(x, w, b, s, p)
    with hip.option('wino', 0):
        got = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), k, k, s, p)
        close(got, want, rtol=2e-5, what='small-map fwd vs torch')
        gy = rnd(tuple(want.shape), 24)
        dx = hip.conv2d_bwd_data(cl(gy), to_tkc(w).to(dev()), (H, W), k, k, s, p)
        close(dx, torch.nn.grad.conv2d_input(x.shape, w, gy, s, p), rtol=2e-5, what='small-map dgrad vs torch')
        with hip.option('debug_now16', 1):
            got0 = hip.conv2d_fwd(cl(x), to_tck(w).to(dev()), b.to(dev()), k, k, s, p)
            dx0 = hip.conv2d_bwd_data(cl(gy), to_tkc(w).to(dev()), (H, W), k, k, s, p)
        close(got, got0, rtol=2e-6, what='split-K vs plain tiles, fwd')
        close(dx, dx0, rtol=2e-6, what='split-K vs plain tiles, dgrad')


@pytest.mark.parametrize('case', [(2, 64, 256, 256, False), (4, 32, 128, 128, True), (16, 64, 64, 64, True), (3, 32, 100, 256, False)],
                         ids=lambda c: 'N%d_C%d_%dx%d_%s' % (c[0], c[1], c[2], c[3], 'dgrad' if c[4] else 'fwd'))
def test_four_cout_kernel_on_bf16_input(mrdis, case):
    """conv3x3_co4_kernel<.., XB>: the window-free 4-cout convolution reading a bf16 map (a lane's 16-byte load = eight channels, widened in
    registers) with the filter in the column-padded [9][C][16] layout -- forward of the C -> 4 layer (MRDIS_DT_XBF16_YF32: bf16 x, fp32 y) and
    data gradient of the 4 -> C si_layers (MRDIS_DT_XF32_YBF16: bf16 dy, fp32 dx, taps reversed): torch fp32 on the same bf16-valued operand
    to fp32 rounding, and the fp32 kernel on the widened input likewise (only the order of the channel sum differs)."""
    hip = mrdis.hip
    N, C, H, W, dgrad = case
    B16 = torch.bfloat16
    if not dgrad:
        x = rnd((N, C, H, W), 31).bfloat16(); w = rnd((4, C, 3, 3), 32, 0.1); b = rnd((4,), 33, 0.1)
        want = F.conv2d(x.float(), w, b, 1, 1)
        tck16 = F.pad(to_tck(w), (0, 12)).contiguous().to(dev())                 # [9][C][16]
        b16 = F.pad(b, (0, 12)).to(dev())
        out = hip.empty_nhwc(N, 4, H, W, dev(), torch.float32)
        got = hip.conv2d_fwd(cl(x.float()).to(B16), tck16, b16, 3, 3, 1, 1, out=out, may_decline=True)
        assert got is not None
        close(got, want, rtol=2e-5, what='C -> 4 forward on bf16 input vs torch')
        ref = hip.conv2d_fwd(cl(x.float()), to_tck(w).to(dev()), b.to(dev()), 3, 3, 1, 1)
        close(got, ref, rtol=2e-6, what='vs the fp32 kernel on the widened input')
    else:
        w = rnd((C, 4, 3, 3), 34, 0.1)                                            # the 4 -> C layer
        dy = rnd((N, C, H, W), 35).bfloat16()
        want = torch.nn.grad.conv2d_input((N, 4, H, W), w, dy.float(), 1, 1)
        tkc16 = F.pad(to_tkc(w), (0, 12)).contiguous().to(dev())                  # [9][C][16]
        out = hip.empty_nhwc(N, 4, H, W, dev(), torch.float32)
        got = hip.conv2d_bwd_data(cl(dy.float()).to(B16), tkc16, (H, W), 3, 3, 1, 1, out=out, may_decline=True)
        assert got is not None
        close(got, want, rtol=2e-5, what='4 <- C data gradient on bf16 dy vs torch')
        ref = hip.conv2d_bwd_data(cl(dy.float()), to_tkc(w).to(dev()), (H, W), 3, 3, 1, 1)
        close(got, ref, rtol=2e-6, what='vs the fp32 kernel on the widened gradient')
