"""Random shapes through the round-5 kernels that multiply on the bf16 matrix pipe with fp32 operands split into bf16 terms, against torch (float64 where the
claim is fp32 accuracy).  Prints the worst error per kernel family; exit code 1 on a mismatch.
  fp32 six-product forms (option split6): c4conv_split6 (4 -> C forward, C <- 4 data gradient), conv3x3_co4<SPLIT> (C -> 4 forward, 4 <- C data gradient),
      conv3x3_c16_split6 (32 -> 16), wgrad16_split6 (32 -> 16), conv3x3_c16t_split6 (16 -> 32 data gradient, split6 = 7)
  bf16-storage forms: conv3x3_co4<XB>, c4conv bf16 output (two terms), wgrad_c4b (4 -> C) and its SWAP form (C -> 4)
    python tools/fuzz_r05.py [trials]"""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402

hip = mrdis.hip
dev = torch.device('cuda:0')
random.seed(11); torch.manual_seed(11)
TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B16 = torch.bfloat16


def cl(t):
    return t.to(dev).contiguous(memory_format=torch.channels_last)


def tck(w):      # (Co, Ci, 3, 3) -> [9][Ci][Co]
    return w.permute(2, 3, 1, 0).reshape(9, w.shape[1], w.shape[0]).contiguous().to(dev)


def tkc(w):      # (Co, Ci, 3, 3) -> [9][Co][Ci]
    return w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).contiguous().to(dev)


def rel(a, b):
    return float((a.detach().double().cpu() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


worst = {}
ok = True


def check(name, err, bar, what):
    global ok
    worst[name] = max(worst.get(name, 0.0), err)
    if not err <= bar:
        print('MISMATCH', name, what, err, '>', bar); ok = False


def big_hw(n, wset=None):
    """a map with n * H * W comfortably above the kernels' size thresholds, odd heights allowed"""
    W = random.choice(wset) if wset else random.randint(33, 300)
    H = max(3, (140000 // (n * W)) + random.randint(0, 9))
    return H, W


for trial in range(TRIALS):
    # ---- fp32, split6: 4 -> C forward (<= 32 couts under the default policy; every width under split6 = 4) and the C <- 4 data gradient
    N = random.choice([1, 2, 3]); Co = random.choice([16, 32, 64, 128]); H, W = big_hw(N)
    x = torch.randn(N, 4, H, W); w = torch.randn(Co, 4, 3, 3) * 0.2; b = torch.randn(Co) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    with hip.option('split6', 4):
        y = hip.conv2d_fwd(cl(x), tck(w), b.to(dev), 3, 3, 1, 1)
    check('c4conv_split6 fwd', rel(y, ref), 2e-6, (N, Co, H, W))
    w2 = torch.randn(4, Co, 3, 3) * 0.2; dy = torch.randn(N, 4, H, W)                      # a Co -> 4 layer: its data gradient is a 4 -> Co convolution
    ref = torch.nn.grad.conv2d_input((N, Co, H, W), w2.double(), dy.double(), 1, 1)
    with hip.option('split6', 4):
        dx = hip.conv2d_bwd_data(cl(dy), tkc(w2), (H, W), 3, 3, 1, 1)
    check('c4conv_split6 dgrad', rel(dx, ref), 2e-6, (N, Co, H, W))
    # ---- fp32, split6: C -> 4 forward, 4 <- C data gradient (W in 64 / 128 / 256)
    C = random.choice([32, 64]); N = random.choice([2, 3, 5]); H, W = big_hw(N, [64, 128, 256])
    x = torch.randn(N, C, H, W); w = torch.randn(4, C, 3, 3) * 0.1; b = torch.randn(4) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    y = hip.conv2d_fwd(cl(x), tck(w), b.to(dev), 3, 3, 1, 1)
    check('co4 split6 fwd', rel(y, ref), 2e-6, (N, C, H, W))
    w2 = torch.randn(C, 4, 3, 3) * 0.1; dy = torch.randn(N, C, H, W)
    ref = torch.nn.grad.conv2d_input((N, 4, H, W), w2.double(), dy.double(), 1, 1)
    dx = hip.conv2d_bwd_data(cl(dy), tkc(w2), (H, W), 3, 3, 1, 1)
    check('co4 split6 dgrad', rel(dx, ref), 2e-6, (N, C, H, W))
    # ---- fp32, split6: 32 -> 16 forward, weight gradient, data gradient (any H, W)
    N = random.choice([1, 2, 4]); H, W = big_hw(N)
    x = torch.randn(N, 32, H, W); w = torch.randn(16, 32, 3, 3) * 0.1; b = torch.randn(16) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    y = hip.conv2d_fwd(cl(x), tck(w), b.to(dev), 3, 3, 1, 1)
    check('c16_split6 fwd', rel(y, ref), 2e-6, (N, H, W))
    dy = torch.randn(N, 16, H, W)
    w64 = torch.zeros(16, 32, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, 1, 1).backward(dy.double())
    dw, db = hip.conv2d_bwd_weight(cl(x), cl(dy), 3, 3, 1, 1, need_bias=True)
    check('wgrad16_split6 dw', rel(dw, tck(w64.grad).cpu()), 2e-6, (N, H, W))
    check('wgrad16_split6 db', rel(db, dy.double().sum((0, 2, 3))), 2e-6, (N, H, W))
    ref = torch.nn.grad.conv2d_input((N, 32, H, W), w.double(), dy.double(), 1, 1)
    with hip.option('split6', 7):
        dx = hip.conv2d_bwd_data(cl(dy), tkc(w), (H, W), 3, 3, 1, 1)
    check('c16t_split6 dgrad', rel(dx, ref), 2e-6, (N, H, W))
    # ---- bf16 storage: C -> 4 forward on a bf16 map / 4 <- C data gradient on a bf16 gradient (three-term filter: exact products)
    C = random.choice([32, 64]); N = random.choice([2, 3, 5]); H, W = big_hw(N, [64, 128, 256])
    xb = torch.randn(N, C, H, W).bfloat16(); w = torch.randn(4, C, 3, 3) * 0.1; b = torch.randn(4) * 0.1
    ref = F.conv2d(xb.double(), w.double(), b.double(), 1, 1)
    out = hip.empty_nhwc(N, 4, H, W, dev, torch.float32)
    y = hip.conv2d_fwd(cl(xb.float()).to(B16), F.pad(tck(w), (0, 12)).contiguous(), F.pad(b, (0, 12)).to(dev), 3, 3, 1, 1, out=out, may_decline=True)
    check('co4 bf16-x fwd', rel(y, ref) if y is not None else 1.0, 2e-6, (N, C, H, W))
    w2 = torch.randn(C, 4, 3, 3) * 0.1; dyb = torch.randn(N, C, H, W).bfloat16()
    ref = torch.nn.grad.conv2d_input((N, 4, H, W), w2.double(), dyb.double(), 1, 1)
    out = hip.empty_nhwc(N, 4, H, W, dev, torch.float32)
    dx = hip.conv2d_bwd_data(cl(dyb.float()).to(B16), F.pad(tkc(w2), (0, 12)).contiguous(), (H, W), 3, 3, 1, 1, out=out, may_decline=True)
    check('co4 bf16-dy dgrad', rel(dx, ref) if dx is not None else 1.0, 2e-6, (N, C, H, W))
    # ---- bf16 storage: 4 -> C forward with a bf16 result (two-term form; error = the bf16 rounding of the result, 2^-8 of the value scale at most)
    Co = random.choice([32, 64, 128]); N = random.choice([1, 2, 3]); H, W = big_hw(N)
    x = torch.randn(N, 4, H, W); w = torch.randn(Co, 4, 3, 3) * 0.2; b = torch.randn(Co) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    y = hip.conv2d_fwd(cl(x), F.pad(tck(w), (0, 0, 0, 12)).contiguous(), b.to(dev), 3, 3, 1, 1, out_dtype=B16, may_decline=True)
    check('c4conv bf16-out fwd', rel(y.float(), ref) if y is not None else 1.0, 4e-3, (N, Co, H, W))
    # ---- bf16 storage: weight gradients of the 4 -> C and C -> 4 layers (three-term fp32 side: exact products)
    Co = random.choice([32, 64, 128]); W = {32: 256, 64: 128, 128: 64}[Co] if random.random() < 0.7 else 64; N = random.choice([2, 3, 9]); H = max(4, 140000 // (N * W) + random.randint(0, 5))
    x = torch.randn(N, 4, H, W); dyb = torch.randn(N, Co, H, W).bfloat16()
    w64 = torch.zeros(Co, 4, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, 1, 1).backward(dyb.double())
    r = hip.conv2d_bwd_weight(cl(x), cl(dyb.float()).to(B16), 3, 3, 1, 1, need_bias=True, may_decline=True)
    if r is not None:
        check('wgrad_c4b dw', rel(r[0], tck(w64.grad).cpu()), 2e-6, (N, Co, H, W))
        check('wgrad_c4b db', rel(r[1], dyb.double().sum((0, 2, 3))), 2e-6, (N, Co, H, W))
    C = random.choice([32, 64]); W = random.choice([64, 128, 256]); N = random.choice([2, 3, 5]); H = max(4, 140000 // (N * W) + random.randint(0, 5))
    xb = torch.randn(N, C, H, W).bfloat16(); dy = torch.randn(N, 4, H, W)
    w64 = torch.zeros(4, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xb.double(), w64, None, 1, 1).backward(dy.double())
    r = hip.conv2d_bwd_weight(cl(xb.float()).to(B16), cl(dy), 3, 3, 1, 1, need_bias=True, may_decline=True)
    if r is not None:
        check('wgrad_c4b SWAP dw', rel(r[0], tck(w64.grad).cpu()), 2e-6, (N, C, H, W))
        check('wgrad_c4b SWAP db', rel(r[1], dy.double().sum((0, 2, 3))), 2e-6, (N, C, H, W))
    torch.cuda.synchronize()
print(f'{TRIALS} random shapes per kernel; worst error relative to the largest reference value (float64 references):')
for k, v in worst.items():
    print(f'  {k:24s} {v:.2e}')
sys.exit(0 if ok else 1)
