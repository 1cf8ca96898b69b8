"""bf16-operand convolution entry points (fwd / dgrad / wgrad, fp32 and bf16 storage) against torch on bf16-rounded operands: error table over a shape list."""
import sys; sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F, mrdis
hip = mrdis.hip; dev = torch.device('cuda:0')
def cl(x): return x.to(dev).contiguous(memory_format=torch.channels_last)
def to_tck(w): Co, Ci, kh, kw = w.shape; return w.permute(2, 3, 1, 0).reshape(kh * kw, Ci, Co).contiguous()
def to_tkc(w): Co, Ci, kh, kw = w.shape; return w.permute(2, 3, 0, 1).reshape(kh * kw, Co, Ci).contiguous()
torch.manual_seed(0)
cases = [(2, 32, 64, 40, 56, 3, 1, 1), (3, 64, 32, 33, 21, 3, 1, 1), (2, 128, 256, 16, 16, 3, 1, 1), (4, 16, 48, 20, 24, 3, 1, 1), (2, 32, 16, 50, 70, 3, 1, 1),
         (2, 32, 64, 32, 48, 4, 2, 1), (8, 128, 128, 8, 8, 3, 1, 1), (2, 64, 128, 24, 24, 3, 2, 1), (2, 512, 128, 16, 16, 3, 1, 1), (1, 16, 16, 7, 9, 1, 1, 0)]
for (N, Ci, Co, H, W, k, s, p) in cases:
    x = torch.randn(N, Ci, H, W); w = torch.randn(Co, Ci, k, k) * (1.0 / (Ci * k * k) ** 0.5); b = torch.randn(Co) * 0.1
    xb, wb = x.bfloat16().float(), w.bfloat16().float()          # what the kernel multiplies
    want = F.conv2d(xb, wb, b, s, p)
    got = hip.conv2d_fwd(cl(x), to_tck(w).to(dev), b.to(dev), k, k, s, p, w_bf16=hip.cast_bf16(to_tkc(w).to(dev)))
    e1 = float((got.cpu() - want).abs().max() / want.abs().max())
    full = F.conv2d(x, w, b, s, p)
    e2 = float((got.cpu() - full).abs().max() / full.abs().max())
    gy = torch.randn_like(want)
    gyb = gy.bfloat16().float()
    want_dx = torch.nn.grad.conv2d_input(x.shape, wb, gyb, s, p)
    dx = hip.conv2d_bwd_data(cl(gy), to_tkc(w).to(dev), (H, W), k, k, s, p, w_bf16=hip.cast_bf16(to_tck(w).to(dev)))
    e3 = float((dx.cpu() - want_dx).abs().max() / want_dx.abs().max())
    gl = hip.conv2d_fwd(cl(x), to_tck(w).to(dev), b.to(dev), k, k, s, p, lrelu=True, w_bf16=hip.cast_bf16(to_tkc(w).to(dev)))
    e4 = float((gl.cpu() - F.leaky_relu(want, 0.2)).abs().max() / want.abs().max())
    print((N, Ci, Co, H, W, k, s, p), f'fwd vs bf16-rounded ref {e1:.2e}, vs fp32 {e2:.2e}; dgrad {e3:.2e}; lrelu {e4:.2e}', flush=True)

print('--- weight gradient')
for (N, Ci, Co, H, W, k) in [(2, 32, 64, 40, 56, 3), (7, 64, 32, 33, 21, 3), (20, 128, 256, 16, 16, 3), (4, 32, 16, 64, 48, 3), (70, 128, 128, 8, 8, 3),
                             (5, 512, 128, 32, 32, 3), (6, 64, 64, 24, 40, 1), (16, 32, 32, 16, 16, 3), (3, 96, 40, 50, 30, 3)]:
    p = (k - 1) // 2
    x = torch.randn(N, Ci, H, W); gy = torch.randn(N, Co, H, W)
    xb, gb = x.bfloat16().float(), gy.bfloat16().float()
    w = torch.zeros(Co, Ci, k, k, requires_grad=True)
    F.conv2d(xb, w, None, 1, p).backward(gb)
    dw, db = hip.conv2d_bwd_weight(cl(x), cl(gy), k, k, 1, p, need_bias=True, dtype=hip.DT_F32_BF16M)
    e1 = float((dw.cpu() - to_tck(w.grad)).abs().max() / w.grad.abs().max())
    e2 = float((db.cpu() - gy.sum((0, 2, 3))).abs().max() / gy.sum((0, 2, 3)).abs().max())
    dw32, _ = hip.conv2d_bwd_weight(cl(x), cl(gy), k, k, 1, p, need_bias=True)
    e3 = float((dw - dw32).abs().max() / dw32.abs().max())
    print((N, Ci, Co, H, W, k), f'wgrad vs bf16-rounded ref {e1:.2e}; dbias {e2:.2e}; vs fp32 kernel {e3:.2e}', flush=True)
