import torch, sys
dev = torch.device('cuda:0')
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mb in (59, 118, 200, 236, 252, 268, 300, 472, 944, 1888):
    n = mb * 1000 * 1000 // 4
    y = torch.empty(n, device=dev); x = torch.randn(n, device=dev)
    tf = t(lambda: y.fill_(1.0)); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.sum())
    print(f'{mb:4d} MB: fill {tf:7.1f} us = {mb / tf * 1e3 / 1e3:5.2f} TB/s | copy {tc:7.1f} us = {2 * mb / tc:5.2f} TB/s (r+w) | sum(read) {tr:7.1f} us = {mb / tr:5.2f} TB/s')
