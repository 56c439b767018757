"""The attention core at the BASELINE model shapes (B = 1, T = 2048, causal), three routes through the registry:
  steps     bmm_0 -> + mask, clamp -> softmax -> bmm_1                 (what the reference's modules run, HIP products)
  folded    bmm_0 -> softmax_bmm (mask / causal / softmax inside the second product)
  one_pass  attention (both products, mask and softmax in one kernel; nothing [heads, T, T] is ever written)
HIP events over 20 calls each.  Run on the GPU box from the repo root."""
import json, math, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import torch.nn.functional as F
import mi355q.quantize as Q

dev = torch.device("cuda:0")
cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
           data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16])


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


for name, H, T, hd, scale in (("OPT-125m", 12, 2048, 64, None), ("OPT-1.3B", 32, 2048, 64, None), ("Llama-7B", 32, 2048, 128, math.sqrt(128)),
                              ("OPT-125m T=1024", 12, 1024, 64, None), ("OPT-125m T=512", 12, 512, 64, None),
                              ("Llama-7B T=1024", 32, 1024, 128, math.sqrt(128)), ("Llama-7B T=512", 32, 512, 128, math.sqrt(128)),
                              ("Llama-2-7B T=4096", 32, 4096, 128, math.sqrt(128))):
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(H, T, hd, generator=g).to(dev) for _ in range(3))
    mask = torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)
    bmm, sbmm, att = Q.get_quantized_func("bmm", cfg), Q.get_quantized_func("softmax_bmm", cfg), Q.get_quantized_func("attention", cfg)

    def steps():
        w = bmm(q, k.transpose(1, 2), config=cfg)
        if scale:
            w = w / scale
        w = torch.max(w + mask, w.new_full((), torch.finfo(w.dtype).min))
        return bmm(F.softmax(w, dim=-1), v, config=cfg)

    def folded():
        w = bmm(q, k.transpose(1, 2), config=cfg)
        if scale:
            w = w / scale
        return sbmm(w, v, cfg, causal=True)

    def one_pass():
        return att(q, k, v, cfg, cfg, causal=True, scale_div=scale)

    ref = steps()
    err = float((one_pass() - ref).abs().max() / ref.abs().max())
    flops = 2 * 2 * H * T * T * hd          # both products, full (unmasked) count
    from mi355q import ops
    r = {"shape": f"{name}: q, k, v [{H}, {T}, {hd}], causal", "steps_us": round(timed(steps), 1), "folded_us": round(timed(folded), 1)}
    for label, which in (("one_pass_resident_us", 1), ("one_pass_resident_8_key_waves_us", 3), ("one_pass_stream_us", 2)):
        if which in (1, 3) and (T > 2048 or (which == 3 and hd > 64)):
            continue
        prev = ops.attention_set_kernel(which)
        try:
            r[label] = round(timed(one_pass), 1)
        finally:
            ops.attention_set_kernel(prev)
    r["one_pass_us"] = round(timed(one_pass), 1)
    prev = ops.attention_set_qpack(2)
    try:
        r["one_pass_q_fragments_packed_us"] = round(timed(one_pass), 1)
    finally:
        ops.attention_set_qpack(prev)
    prev = ops.attention_set_qpack(0)                                  # (round 6: Q fragments packed in front of the kernels, or not)
    try:
        r["one_pass_q_quantised_in_kernel_us"] = round(timed(one_pass), 1)
    finally:
        ops.attention_set_qpack(prev)
    r["one_pass_vs_steps_max_rel"] = round(err, 6)
    r["speedup_vs_steps"] = round(r["steps_us"] / r["one_pass_us"], 2)
    r["one_pass_TFLOPs_full_count"] = round(flops / r["one_pass_us"] / 1e6, 1)
    print(json.dumps(r), flush=True)
