import torch
dev = torch.device("cuda:0")
a = torch.randint(-31, 32, (4096, 4096), dtype=torch.int8, device=dev); b = torch.randint(-31, 32, (4096, 4096), dtype=torch.int8, device=dev)
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
try:
    us = t(lambda: torch._int_mm(a, b.t()))
    print("torch._int_mm 4096^3 int8 -> int32:", round(us, 1), "us", round(2 * 4096 ** 3 / us / 1e6), "TOPS")
except Exception as ex:
    print("int_mm failed:", type(ex).__name__, str(ex)[:200])
x = torch.randn(4096, 4096, device=dev).bfloat16(); w = torch.randn(4096, 4096, device=dev).bfloat16()
us = t(lambda: torch.mm(x, w.t(), out_dtype=torch.float32)); print("bf16 -> fp32 mm:", round(us, 1), "us", round(2 * 4096 ** 3 / us / 1e6), "TFLOP/s")
us = t(lambda: torch.mm(x, w.t())); print("bf16 -> bf16 mm:", round(us, 1), "us", round(2 * 4096 ** 3 / us / 1e6), "TFLOP/s")
