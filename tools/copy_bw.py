import torch, statistics
dev = torch.device("cuda:0")
x = torch.randn(131072, 768, device=dev); y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        a,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e)*1e3)
    return statistics.median(ts)
us = t(lambda: y.copy_(x)); print("copy 403MB->403MB: %.1f us = %.2f TB/s" % (us, 2*x.numel()*4/us/1e6))
us = t(lambda: torch.add(x, 1.0, out=y)); print("add  : %.1f us = %.2f TB/s" % (us, 2*x.numel()*4/us/1e6))
us = t(lambda: x.sum()); print("sum  : %.1f us = %.2f TB/s (read only)" % (us, x.numel()*4/us/1e6))
