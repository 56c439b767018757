"""Race screen for the attention kernels: the same call repeated must give bit-identical results (cross-wave reductions
through LDS, LDS-DMA staging and barriers of the streaming kernel)."""
import sys, math
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
from mi355q import ops
dev = torch.device("cuda:0")
cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
           data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16])
att = Q.get_quantized_func("attention", cfg)
for (H, T, hd, which, reps) in [(12, 2048, 64, 1, 150), (12, 2048, 64, 3, 150), (12, 2048, 64, 2, 100), (32, 2048, 128, 1, 60),
                                (32, 2048, 128, 2, 40), (8, 1008, 96, 1, 150), (8, 1008, 96, 2, 150), (4, 4096, 128, 2, 40)]:
    g = torch.Generator().manual_seed(H + T + hd)
    q, k, v = (torch.randn(H, T, hd, generator=g).to(dev) for _ in range(3))
    prev = ops.attention_set_kernel(which if T <= 2048 else 0)
    try:
        ref = att(q, k, v, cfg, cfg, causal=True, scale_div=math.sqrt(hd) if hd == 128 else None).clone()
        bad = sum(int(not torch.equal(ref, att(q, k, v, cfg, cfg, causal=True, scale_div=math.sqrt(hd) if hd == 128 else None)))
                  for _ in range(reps))
    finally:
        ops.attention_set_kernel(prev)
    print(f"attention [{H}, {T}, {hd}] kernel {which}: {reps} runs, {bad} differ", flush=True)
