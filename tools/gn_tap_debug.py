import sys, torch
sys.path.insert(0, '/root/repo')
import mrdis
from mrdis import hip
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
N, C, D, H, W = 2, 16, 12, 20, 24
def cl3(t): return t.to(dev).contiguous(memory_format=torch.channels_last_3d)
x = cl3(torch.randn(N, C, D, H, W, generator=g)); dy = cl3(torch.randn(N, C, D, H, W, generator=g)); add = cl3(torch.randn(N, C, D, H, W, generator=g))
gamma = torch.randn(C, generator=g).to(dev); beta = torch.randn(C, generator=g).to(dev)
y, mean, rstd = hip.groupnorm_relu_fwd(x, gamma, beta, 8, 1e-5, True)
dx0, dg0, db0 = hip.groupnorm_relu_bwd(dy, x, gamma, beta, mean, rstd, 8, True)
dx1, dg1, db1 = hip.groupnorm_relu_bwd(dy, x, gamma, beta, mean, rstd, 8, True, add=add)
ref = dx0 + add
print('kernel add vs torch add: equal', torch.equal(ref, dx1), 'max diff', float((ref - dx1).abs().max()), 'dg equal', torch.equal(dg0, dg1))
m3 = mrdis.model3d
torch.manual_seed(3)
blk = mrdis.BasicBlock(16, 16).to(dev)
res = {}
for tap in (False, True):
    m3._GN_TAP = tap
    xg = x.clone().requires_grad_(True)
    pre = xg * 1.0
    yb = blk(pre)
    yb.backward(dy)
    res[tap] = (xg.grad.clone(), [p.grad.clone() for p in blk.parameters()])
    blk.zero_grad()
print('xg.grad equal', torch.equal(res[False][0], res[True][0]), float((res[False][0] - res[True][0]).abs().max()), float(res[False][0].abs().max()))
for (n, _), a, b in zip(blk.named_parameters(), res[False][1], res[True][1]):
    print(n, torch.equal(a, b), float((a - b).abs().max()))
