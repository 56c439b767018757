"""softmax folded into the quantise + P V product (mi355q_bfp_softmax_matmul) against softmax, then the product."""
import json, sys; sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
dev = torch.device('cuda:0')
cfg = dict(name="block_fp", data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16])
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
for name, (h, T, hd) in (("OPT-125m", (12, 2048, 64)), ("OPT-1.3B", (32, 2048, 64)), ("Llama-7B", (32, 2048, 128))):
    s = torch.randn(h, T, T, device=dev) * 3 + torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)
    v = torch.randn(h, T, hd, device=dev)
    bmm, sbmm = Q.get_quantized_func("bmm", cfg), Q.get_quantized_func("softmax_bmm", cfg)
    t3 = t(lambda: bmm(torch.softmax(s, -1), v, config=cfg))
    tsm = t(lambda: torch.softmax(s, -1))
    t1 = t(lambda: sbmm(s, v, config=cfg))
    nbytes = s.numel() * 4
    print(json.dumps({"shape": name, "heads": h, "T": T, "head_dim": hd, "softmax_then_product_us": round(t3, 1), "of_which_softmax_us": round(tsm, 1),
                      "folded_us": round(t1, 1), "speedup": round(t3 / t1, 2), "probability_tensor_MB_not_written_and_not_reread": round(2 * nbytes / 1e6, 1),
                      "folded_TBps_of_scores": round(nbytes / t1 / 1e6, 2)}))
