"""One Linear layer, post-activation input, through every alignment mode of the module (registry API, torch events)."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device("cuda:0")
def cfg(align):
    return dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
                data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
                weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
                mi355q_align=align)
torch.manual_seed(0)
for name, (M, K, N), act in (("OPT-125m fc2 (ReLU)", (2048, 3072, 768), "relu"), ("OPT-1.3B fc2 (ReLU)", (2048, 8192, 2048), "relu"),
                             ("Llama-7B down (SiLU*up)", (2048, 11008, 4096), "silu"), ("plain 4096^3", (4096, 4096, 4096), "none")):
    h = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
    x = {"relu": torch.relu(h), "silu": torch.nn.functional.silu(h) * torch.randn(M, K, device=dev), "none": h}[act]
    fp = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad(): fp.weight.normal_(0, 0.02)
    out = []
    for align in ("auto", "rows", "rows_post", "blocks"):
        lin = Q.get_quantized_cls("linear", cfg(align)).from_float(fp, cfg(align)).to(dev)
        with torch.no_grad():
            for _ in range(5): lin(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): lin(x)
            b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        out.append(f"{align}{'(' + str(lin._x_cap) + ')' if align == 'auto' else ''} {us:.0f}")
    print(f"{name:26s} M={M} K={K} N={N}: " + " | ".join(out) + "  us")
