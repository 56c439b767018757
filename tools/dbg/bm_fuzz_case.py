import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import mi355q.quantize as Q
from mi355q import ops
dev = torch.device("cuda:0")
cfg = dict(name="block_minifloat", bypass=False, data_in_width=8, data_in_exponent_width=4, data_in_exponent_bias_width=8, data_in_block_size=[1, 16],
           weight_width=8, weight_exponent_width=4, weight_exponent_bias_width=8, weight_block_size=[1, 16])
r = np.random.default_rng(5)
found = 0
for it in range(400):
    B, M, K, N = int(r.integers(1, 5)), int(r.integers(1, 150)), 16 * int(r.integers(1, 20)), 16 * int(r.integers(1, 12))
    sx, sy = 10.0 ** r.integers(-3, 4), 10.0 ** r.integers(-2, 3)
    x = (r.normal(size=(B, M, K)) * sx).astype(np.float32)
    y = (r.normal(size=(B, K, N)) * sy).astype(np.float32)
    xt, yt = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    xq16 = ops.block_minifloat_quantize_bf16(xt, 8, 4, 8, [1, 16], True).float()
    xq32 = ops.block_minifloat_quantize(xt, 8, 4, 8, [1, 16], True)
    yq16 = ops.block_minifloat_quantize_bf16(yt, 8, 4, 8, [1, 16], True).float()
    yq32 = ops.block_minifloat_quantize(yt, 8, 4, 8, [1, 16], True)
    dx, dy = (xq16 - xq32).abs().max().item(), (yq16 - yq32).abs().max().item()
    a = torch.bmm(xq16.bfloat16(), yq16.bfloat16(), out_dtype=torch.float32)
    b = torch.bmm(xq32, yq32)
    c = torch.bmm(xq32.double(), yq32.double()).float()
    e_ab = ((a - c).abs().max() / (c.abs().max() + 1e-30)).item()
    e_b = ((b - c).abs().max() / (c.abs().max() + 1e-30)).item()
    if dx or dy or e_ab > 3e-6:
        found += 1
        print(it, (B, M, K, N), "scales", sx, sy, "operand diffs", dx, dy, "bf16 product vs exact", e_ab, "fp32 product vs exact", e_b,
              "max |xq|", xq32.abs().max().item(), "max |yq|", yq32.abs().max().item())
        if found > 6: break
print("found", found)
