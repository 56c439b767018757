import torch
dev = torch.device('cuda:0')
y = torch.empty(4096, 4096, device=dev)
x = torch.randn(4096, 4096, device=dev)
for name, fn in (("zero_ 64MB", lambda: y.zero_()), ("copy_ 64MB->64MB", lambda: y.copy_(x)), ("mul_ inplace", lambda: y.mul_(1.5))):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 50 * 1e3, "us")
