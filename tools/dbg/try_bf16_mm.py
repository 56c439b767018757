import sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
dev = torch.device("cuda:0")
for (M, N, K) in ((2048, 768, 2048), (2048, 4096, 11008), (4096, 4096, 4096)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02
    xq = ops.block_fp_quantize(x, 6, 8, 127, [1, 16], True); wq = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False)
    xb, wb = xq.bfloat16(), wq.bfloat16()
    print("non-representable elements:", int((xb.float() != xq).sum()), int((wb.float() != wq).sum()))
    try:
        y = torch.mm(xb, wb.t(), out_dtype=torch.float32)
    except Exception as e:
        print("mm out_dtype failed:", type(e).__name__, str(e)[:150]); break
    ref = (xq.double() @ wq.double().t())
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    def t(f, n=20):
        for _ in range(3): f()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): f()
        b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
    t_mm = t(lambda: torch.mm(xb, wb.t(), out_dtype=torch.float32))
    t_q = t(lambda: ops.block_fp_quantize(x, 6, 8, 127, [1, 16], True).bfloat16())
    t_f32 = t(lambda: xq @ wq.t())
    t_qb = t(lambda: ops.block_fp_quantize_bf16(x, 6, 8, 127, [1, 16], True))
    print(M, N, K, "bf16 mm->fp32", round(t_mm, 1), "us", round(2 * M * N * K / t_mm / 1e6), "TFLOP/s | quant+cast", round(t_q, 1), "us | quant->bf16 direct", round(t_qb, 1), "us | fp32 mm", round(t_f32, 1), "us | rel err", err)
