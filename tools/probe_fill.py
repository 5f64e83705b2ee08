import torch, time
dev = torch.device("cuda:0")
n = 16 * 1024**3
x = torch.empty(n, dtype=torch.uint8, device=dev)
y = torch.empty(n // 2, dtype=torch.uint8, device=dev)
def t(fn, name, bytes_):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"{name}: {ms:.3f} ms {bytes_/ms/1e6:.0f} GB/s", flush=True)
t(lambda: x.zero_(), "zero_ 16GB", n)
t(lambda: x.fill_(7), "fill_ 16GB", n)
xi = x.view(torch.int32)
t(lambda: xi.fill_(7), "fill_ int32 16GB", n)
t(lambda: x[:n//2].copy_(y), "copy 8GB->8GB (r+w)", n)
