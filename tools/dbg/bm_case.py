import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
import mi355q.quantize as Q
from mi355q import ops
cfg = dict(name="block_minifloat", bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8, data_in_block_size=[1, 16],
           weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8, weight_block_size=[1, 16])
g = torch.Generator().manual_seed(0)
f = Q.get_quantized_func("matmul", cfg)
for scale in (1.0, 100.0, 1e4, 1e6):
    for shape in ((2, 4, 48, 64), (2, 4, 48, 48)):
        x = (torch.randn(*shape, generator=g) * scale).cuda()
        y = (torch.randn(2, 4, shape[-1], 64, generator=g) * scale).cuda()
        a = f(x, y, dict(cfg)); b = f(x, y, dict(cfg, mi355q_values_matmul="fp32"))
        xq = ops.block_minifloat_quantize_bf16(x.flatten(0, 1), 8, 4, 8, [1, 16], True).float()
        xr = ops.block_minifloat_quantize(x.flatten(0, 1), 8, 4, 8, [1, 16], True)
        print(scale, shape, float((a - b).abs().max() / b.abs().max()), float((xq - xr).abs().max()), float(xr.abs().max()))
# K = 48 (not a multiple of 16? it is) ; T = 48 scores with -inf-like masks
p = torch.softmax(torch.randn(2, 4, 48, 48, generator=g).cuda() * 50 + torch.full((48, 48), float("-inf"), device="cuda").triu(1), -1)
v = torch.randn(2, 4, 48, 64, generator=g).cuda() * 100
a = f(p, v, dict(cfg)); b = f(p, v, dict(cfg, mi355q_values_matmul="fp32"))
print("probs", float((a - b).abs().max() / b.abs().max()))
