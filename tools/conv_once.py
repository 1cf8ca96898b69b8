"""Launch mrdis_conv2d_fwd (3x3 s1 p1) a few times on one shape: target for PMC passes.
    python tools/conv_once.py N Ci H W Co [iters] [img]        img: with the filter image of the library's policy (F(2x2) or F(4x4), option wino4)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mrdis  # noqa: E402
N, Ci, H, W, Co = [int(v) for v in sys.argv[1:6]]
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device('cuda:0')
x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(9, Ci, Co, device=dev) * 0.1
b = torch.randn(Co, device=dev)
img = None
if len(sys.argv) > 7 and sys.argv[7] == 'img':
    from tools.wino4_check import images
    img, _ = images(w, w.permute(0, 2, 1).contiguous(), dev)
for _ in range(iters):
    y = mrdis.hip.conv2d_fwd(x, w, b, 3, 3, 1, 1, w_wino=img)
torch.cuda.synchronize()
