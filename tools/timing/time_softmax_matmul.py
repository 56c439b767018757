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
    raw = torch.randn(h, T, T, device=dev) * 3                       # what the first product (Q K^T) leaves
    mask = torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)
    fmin = torch.tensor(torch.finfo(torch.float32).min, device=dev)
    v = torch.randn(h, T, hd, device=dev)
    bmm, sbmm = Q.get_quantized_func("bmm", cfg), Q.get_quantized_func("softmax_bmm", cfg)
    # the reference's steps (modeling_opt.py:262-312): add the mask, clamp, softmax, product
    t_ref = t(lambda: bmm(torch.softmax(torch.max(raw + mask, fmin), -1), v, config=cfg))
    t_glue = t(lambda: torch.softmax(torch.max(raw + mask, fmin), -1))
    t_causal = t(lambda: sbmm(raw, v, config=cfg, causal=True))
    t_mask = t(lambda: sbmm(raw, v, config=cfg, mask=mask))
    nbytes = raw.numel() * 4
    print(json.dumps({"shape": name, "heads": h, "T": T, "head_dim": hd, "reference_steps_us": round(t_ref, 1), "of_which_mask_clamp_softmax_us": round(t_glue, 1),
                      "folded_causal_us": round(t_causal, 1), "folded_mask_tensor_us": round(t_mask, 1), "speedup_causal": round(t_ref / t_causal, 2),
                      "speedup_mask_tensor": round(t_ref / t_mask, 2), "intermediate_tensors_MB_not_written_and_not_reread": round(6 * nbytes / 1e6, 1)}))
