"""Attention-shaped block_fp products through the registry function: fused quantise + matmul vs two quantisers + GEMM."""
import sys, json
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
dev = torch.device("cuda:0")
def cfg(fused):
    return dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
                data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
                weight_block_size=[1, 16], mi355q_fused_matmul=fused)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
for name, xs, ys in (("OPT-125m probs x V", (12, 2048, 2048), (12, 2048, 64)), ("OPT-125m Q x K^T", (12, 2048, 64), (12, 64, 2048)),
                     ("Llama-7B probs x V", (32, 2048, 2048), (32, 2048, 128)), ("Llama-7B Q x K^T", (32, 2048, 128), (32, 128, 2048))):
    x = torch.randn(xs, device=dev)
    if xs[-1] == 2048:
        x = torch.softmax(x + torch.full((2048, 2048), float("-inf"), device=dev).triu(1), dim=-1)
    y = torch.randn(ys, device=dev)
    f = Q.get_quantized_func("bmm", cfg(True))
    t1 = timeit(lambda: f(x, y, cfg(True)))
    t0 = timeit(lambda: f(x, y, cfg(False)))
    d = (f(x, y, cfg(True)) - f(x, y, cfg(False))).abs().max().item()
    byts = 4 * (x.numel() + y.numel() + xs[0] * xs[1] * ys[2])
    print(json.dumps({"case": name, "x": list(xs), "y": list(ys), "fused_us": round(t1, 1), "two_step_us": round(t0, 1),
                      "max_abs_diff": d, "algorithmic_MB": round(byts / 1e6, 1), "fused_TBps": round(byts / t1 / 1e6, 2)}))
