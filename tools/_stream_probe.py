import torch, time
dev = torch.device('cuda:0')
x = torch.randn(2, 80, 200, 200, device=dev); m = torch.rand(2, 1, 200, 200, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
out = torch.empty_like(x)
print('torch.mul(x, m, out=out)  %.1f us' % t(lambda: torch.mul(x, m, out=out)))
print('out.copy_(x)              %.1f us' % t(lambda: out.copy_(x)))
print('x.sum()                   %.1f us' % t(lambda: x.sum()))
big = torch.randn(64, 80, 200, 200, device=dev); bo = torch.empty_like(big)
print('copy 819 MB               %.1f us -> %.2f TB/s' % ((lambda u: (u, 2 * big.numel() * 4 / u / 1e6))(t(lambda: bo.copy_(big), 20))))
