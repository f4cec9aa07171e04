"""Does a tensor WRITTEN by one kernel come back from the Infinity Cache (256 MB) when the next kernel reads it?
write N bytes (fill), then read them (sum); also copy -> read.  Reports the read kernel's bandwidth per size."""
import torch, time
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        fn_pre()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts)//2]
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16)
    b = torch.empty(n, device=dev, dtype=torch.bfloat16)
    big = torch.empty(1536 * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16)
    # (1) producer wrote it just before
    fn_pre = lambda: a.fill_(1.0)
    ms_hot = t(lambda: b.copy_(a))
    # (2) cache flushed by a 1.5 GB fill between producer and consumer
    def pre2():
        a.fill_(1.0); big.fill_(0.0)
    fn_pre = pre2
    ms_cold = t(lambda: b.copy_(a))
    print(f"{mb:5d} MB: copy right after the producer {2*mb/1024/ms_hot*1e3:6.2f} TB/s (r+w) | after a 1.5 GB flush {2*mb/1024/ms_cold*1e3:6.2f} TB/s", flush=True)
    del a, b, big
