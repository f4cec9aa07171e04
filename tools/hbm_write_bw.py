import torch, time
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (32, 128, 512, 2048):
    x = torch.empty(mb << 20, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
    w = t(lambda: x.zero_()); c = t(lambda: y.copy_(x)); 
    s = t(lambda: x.sum(dtype=torch.int64)) if mb <= 512 else 0
    print(f"{mb:5d} MB: fill {mb/1024/w/1e3:6.2f} TB/s   copy {2*mb/1024/c/1e3:6.2f} TB/s (r+w)")
