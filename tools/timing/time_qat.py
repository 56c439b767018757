"""QAT (is_ptq = False) forward + backward of one LinearBlockFP on the bf16 tile GEMM (own kernels: 1 + 4 launches) against the
library route (F.linear: fp32 GEMM forward and two backward), HIP events.  python tools/timing/time_qat.py [M K N]"""
import json, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
dev = torch.device("cuda:0")
shapes = [tuple(int(v) for v in sys.argv[1:4])] if len(sys.argv) > 3 else [(512, 1024, 4096), (2048, 1024, 4096), (2048, 4096, 4096), (2048, 4096, 11008)]
base = dict(name="block_fp", is_ptq=False, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
            weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
            bias_exponent_bias=127, bias_block_size=[16])
for M, K, N in shapes:
    row = {"M": M, "K": K, "N": N}
    for route in ("bf16", "fp32"):
        cfg = dict(base, mi355q_qat_gemm="bf16_always" if route == "bf16" else route)
        torch.manual_seed(0)
        lin = Q.get_quantized_cls("linear", cfg)(K, N, config=cfg).to(dev)
        x = torch.randn(M, K, device=dev, requires_grad=True)
        dy = torch.randn(M, N, device=dev)
        def step():
            lin.zero_grad(set_to_none=True); x.grad = None
            lin(x).backward(dy)
        for _ in range(5): step()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): step()
        e.record(); torch.cuda.synchronize()
        row[route + "_us"] = round(a.elapsed_time(e) / 20 * 1e3, 1)
    row["speedup"] = round(row["fp32_us"] / row["bf16_us"], 2)
    print(json.dumps(row), flush=True)
