"""PTQ LinearBlockMinifloat / LinearBlockLog at the Llama-7B shapes: the bf16 flavour of the tile GEMM on the fake-quantised
values against the library fp32 GEMM the reference's F.linear maps to (config["mi355q_values_gemm"] = "fp32")."""
import json, sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
dev = "cuda:0"
CFGS = {"block_minifloat": dict(name="block_minifloat", is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_width=4,
                                data_in_exponent_bias_width=8, data_in_block_size=[1, 16], weight_width=8, weight_exponent_width=4,
                                weight_exponent_bias_width=8, weight_block_size=[1, 16], bias_width=8, bias_exponent_width=4,
                                bias_exponent_bias_width=8, bias_block_size=[16]),
        "block_log": dict(name="block_log", is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_bias_width=8,
                          data_in_block_size=[1, 16], weight_width=8, weight_exponent_bias_width=8, weight_block_size=[1, 16],
                          bias_width=8, bias_exponent_bias_width=8, bias_block_size=[16])}
def timed(fn, n=20):
    t_end = time.time() + 0.1
    while time.time() < t_end:
        fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
for arith, cfg in CFGS.items():
    for (M, K, N) in [(2048, 4096, 4096), (2048, 4096, 11008), (2048, 11008, 4096)]:
        torch.manual_seed(0)
        x = torch.randn(M, K, device=dev) * 8
        out = {"layer": f"Linear{arith} {K} -> {N}, {M} tokens"}
        for mode in ("bf16", "fp32"):
            lin = Q.get_quantized_cls("linear", cfg)(K, N, bias=False, config=dict(cfg, mi355q_values_gemm=mode)).to(dev)
            with torch.no_grad():
                lin.weight.mul_(200.0)
                lin(x)
                out[f"{mode}_us"] = round(timed(lambda: lin(x)), 1)
        out["speedup"] = round(out["fp32_us"] / out["bf16_us"], 2)
        print(json.dumps(out), flush=True)

# two decoder layers + head at the Llama-7B width under each arithmetic (attention stepped as the reference does it)
from mi355q.harness import TinyLlamaConfig, TinyLlamaForCausalLM, expand_llama_quant_config
for arith, cfg in CFGS.items():
    out = {"model": f"Llama-7B width, 2 layers, T=2048, {arith} W8"}
    for mode in ("bf16", "fp32"):
        torch.manual_seed(0)
        mc = TinyLlamaConfig(vocab_size=2048, hidden_size=4096, intermediate_size=11008, num_layers=2, num_heads=32, max_positions=2048)
        model = TinyLlamaForCausalLM(mc, expand_llama_quant_config(dict(cfg, mi355q_values_gemm=mode), mc.num_layers)).to(dev).eval()
        ids = torch.randint(0, mc.vocab_size, (1, 2048)).to(dev)
        with torch.no_grad():
            model(ids)
            out[f"{mode}_ms"] = round(timed(lambda: model(ids), n=5) / 1e3, 3)
        del model
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)
