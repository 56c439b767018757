import sys, math
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
dev = torch.device("cuda:0")
cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
           data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16])
for H, T, hd in ((12, 2048, 64), (32, 2048, 128)):
    q, k, v = (torch.randn(H, T, hd, device=dev) for _ in range(3))
    mask = torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)
    att = Q.get_quantized_func("attention", cfg)
    for name, kw in (("causal", dict(causal=True)), ("mask tensor", dict(mask=mask))):
        for _ in range(3): att(q, k, v, cfg, cfg, **kw)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): att(q, k, v, cfg, cfg, **kw)
        e.record(); torch.cuda.synchronize()
        print(H, T, hd, name, round(a.elapsed_time(e) / 20 * 1e3, 1), "us")
