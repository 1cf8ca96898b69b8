"""The six-product tap-table kernel (csrc/mrdis_s6conv.hip: fp32 operands as three bf16 terms on the bf16 matrix pipe, filter image from
mrdis_s6_filter_image, w_wino_fmt = 6) against torch float64 -- the op is F.conv2d / its data gradient (model.py:2104) -- forward and data gradient, strides
1 and 2, every wave tile and channel chunk the launcher knows, ragged maps and channel counts, bias + leaky-ReLU epilogue, views with a leading dimension;
and the path the training step takes it on: the data gradient of a 4x4 stride-2 layer through ops (ops.s6_dgrad_image)."""
import pytest
import torch
import torch.nn.functional as F

DEV = torch.device('cuda:0')
BAR = 2e-6          # of the result's maximum: the bar of the other six-product kernels (fp32 accumulation of K = 16 x 256 terms alone reaches 1.6e-6)


def cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


def tck(w):
    return w.permute(2, 3, 1, 0).reshape(-1, w.shape[1], w.shape[0]).contiguous().to(DEV)


def tkc(w):
    return w.permute(2, 3, 0, 1).reshape(-1, w.shape[0], w.shape[1]).contiguous().to(DEV)


def rel(a, b):
    return float((a.detach().double().cpu() - b.double().cpu()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


CASES = [      # N, Ci, Co, k, stride, pad, H, W
    (4, 32, 64, 4, 2, 1, 64, 64), (2, 64, 128, 4, 2, 1, 32, 48), (3, 128, 256, 4, 2, 1, 16, 16), (4, 16, 32, 3, 2, 1, 64, 64), (2, 32, 64, 3, 2, 1, 33, 47),
    (2, 128, 128, 3, 1, 1, 16, 16), (5, 24, 20, 3, 1, 1, 19, 23), (3, 40, 36, 4, 2, 1, 22, 26), (1, 64, 48, 3, 1, 1, 70, 33),
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=str)
def test_six_product_tap_kernel_forward_and_data_gradient(mrdis, case):
    hip = mrdis.hip
    N, Ci, Co, k, st, pad, H, W = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g); w = torch.randn(Co, Ci, k, k, generator=g) * (Ci * k * k) ** -0.5; b = torch.randn(Co, generator=g) * 0.1
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    wt, wk = tck(w), tkc(w)
    imf, imd = hip.s6_filter_image(wt), hip.s6_filter_image(wk)
    assert imf is not None and imf.numel() * 4 == hip.s6_filter_image_bytes(k * k, Ci, Co)
    assert (imd is None) == (Co % 8 != 0)              # (the data gradient reduces over Co: no image unless it is a multiple of 8 -- the fp32 kernels then)
    ref_y = F.conv2d(x.double(), w.double(), b.double(), st, pad)
    ref_l = F.leaky_relu(ref_y, 0.2)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, W), w.double(), dy.double(), st, pad)
    with hip.option('split6', 10):                     # only this kernel of the six-product family
        hip.launch_counts(reset=True)
        y = hip.conv2d_fwd(cl(x), wt, b.to(DEV), k, k, st, pad, w_wino=imf)
        yl = hip.conv2d_fwd(cl(x), wt, b.to(DEV), k, k, st, pad, lrelu=True, w_wino=imf)
        dx = hip.conv2d_bwd_data(cl(dy), wk, (H, W), k, k, st, pad, w_wino=imd)
        took = hip.launch_counts()['split6_tap']
    assert took >= (3 if imd is not None else 2), took          # (a stride-2 data gradient may need one launch per parity class)
    assert rel(y, ref_y) <= BAR and rel(yl, ref_l) <= BAR and rel(dx, ref_dx) <= BAR, (rel(y, ref_y), rel(yl, ref_l), rel(dx, ref_dx))
    with hip.option('split6', 0):                      # the image is ignored: the fp32 MFMA kernels, same results to rounding
        hip.launch_counts(reset=True)
        y0 = hip.conv2d_fwd(cl(x), wt, b.to(DEV), k, k, st, pad, w_wino=imf)
        assert hip.launch_counts()['split6_tap'] == 0
    assert rel(y0, ref_y) <= BAR


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [812, 822, 821, 811, 1612, 1622, 1621, 1611, 3212, 3222, 3221, 3211])
def test_every_wave_tile_and_channel_chunk(mrdis, mode):
    """debug_mode = 100 kc + 10 wp + wc forces one instantiation (declined where the tile does not fit: then the fp32 kernel answers, also within the bar)"""
    hip = mrdis.hip
    N, Ci, Co, H = 8, 64, 64, 48
    g = torch.Generator().manual_seed(mode)
    x = torch.randn(N, Ci, H, H, generator=g); w = torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5
    dy = torch.randn(N, Co, H // 2, H // 2, generator=g)
    wt, wk = tck(w), tkc(w)
    ref_y = F.conv2d(x.double(), w.double(), None, 2, 1)
    ref_dx = torch.nn.grad.conv2d_input((N, Ci, H, H), w.double(), dy.double(), 2, 1)
    with hip.option('split6', 10), hip.option('debug_mode', mode):
        y = hip.conv2d_fwd(cl(x), wt, None, 3, 3, 2, 1, w_wino=hip.s6_filter_image(wt))
        dx = hip.conv2d_bwd_data(cl(dy), wk, (H, H), 3, 3, 2, 1, w_wino=hip.s6_filter_image(wk))
    assert rel(y, ref_y) <= BAR and rel(dx, ref_dx) <= BAR


@pytest.mark.gpu
def test_views_with_a_leading_dimension_and_wrong_images(mrdis):
    hip = mrdis.hip
    N, Ci, Co, H = 4, 32, 64, 64
    g = torch.Generator().manual_seed(3)
    big = cl(torch.randn(N, Ci + 16, H, H, generator=g))
    x = big[:, 8:8 + Ci]                                # channel slice of a wider NHWC buffer: ld = 48, 32-byte offset
    w = torch.randn(Co, Ci, 4, 4, generator=g) * 0.05
    wt = tck(w)
    out_big = hip.empty_nhwc(N, Co + 32, H // 2, H // 2, DEV)
    with hip.option('split6', 10):
        hip.launch_counts(reset=True)
        y = hip.conv2d_fwd(x, wt, None, 4, 4, 2, 1, out=out_big[:, 32:], w_wino=hip.s6_filter_image(wt))
        assert hip.launch_counts()['split6_tap'] == 1
    assert rel(y, F.conv2d(x.double().cpu(), w.double(), None, 2, 1)) <= BAR
    with pytest.raises(mrdis.hip.MrdisError):          # an image of another filter shape is refused by its size, not read
        hip.conv2d_fwd(x, wt, None, 4, 4, 2, 1, w_wino=hip.s6_filter_image(tck(torch.randn(Co, Ci + 8, 4, 4))))
    assert hip.s6_filter_image(tck(torch.randn(16, 7, 4, 4))) is None          # reduction axis not a multiple of 8: no image, the fp32 kernels take the layer


@pytest.mark.gpu
def test_training_path_takes_the_data_gradient_of_the_4x4_stride_2_layers(mrdis):
    """ops.s6_dgrad_image: one image per mixed kernel and step scope, only for 4x4 stride-2 layers on maps of >= 32k positions; gradients as autograd's"""
    hip, ops = mrdis.hip, mrdis.ops
    N, Ci, Co, H = 8, 32, 64, 64
    g = torch.Generator().manual_seed(9)
    x = cl(torch.randn(N, Ci, H, H, generator=g)).requires_grad_(True)
    w = torch.randn(Co, Ci, 4, 4, generator=g) * 0.05
    wt, wk = tck(w), tkc(w)
    dy = torch.randn(N, Co, H // 2, H // 2, generator=g)
    ref = torch.nn.grad.conv2d_input((N, Ci, H, H), w.double(), dy.double(), 2, 1)
    with ops.mix_cache():
        hip.launch_counts(reset=True)
        for _ in range(2):                              # two uses of the same mixed kernel inside a step: one image
            y = torch.ops.mrdis.conv2d(x, wt, wk, None, 4, 4, 2, 1, False, None, None)
            (dx,) = torch.autograd.grad(y, x, cl(dy))
        c = hip.launch_counts()
    assert c['split6_tap'] == 2 and rel(dx, ref) <= BAR, (c['split6_tap'], rel(dx, ref))
    small = cl(torch.randn(2, Ci, 32, 32, generator=g)).requires_grad_(True)          # 2,048 positions: stays on the fp32 kernel
    hip.launch_counts(reset=True)
    ys = torch.ops.mrdis.conv2d(small, wt, wk, None, 4, 4, 2, 1, False, None, None)
    torch.autograd.grad(ys, small, torch.ones_like(ys))
    assert hip.launch_counts()['split6_tap'] == 0
