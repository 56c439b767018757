import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
import mi355q.quantize as Q
from mi355q import ops
dev = torch.device("cuda:0")
base = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
            bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16], mi355q_align="auto")
g = lambda s: torch.Generator().manual_seed(s)
K, N, M = 11008, 4096, 2048
torch.manual_seed(0)
fp = torch.nn.Linear(K, N)
with torch.no_grad():
    fp.weight.mul_(1.5)
x = (torch.nn.functional.silu(torch.randn(M, K, generator=g(0)) * torch.exp(torch.randn(M, 1, generator=g(1)))) * torch.randn(M, K, generator=g(2))).to(dev)
outs = {}
for tag, cfg in (("resident", dict(base)), ("packed", dict(base, mi355q_weight_storage="packed"))):
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
    with torch.no_grad():
        for _ in range(3):
            y = lin(x)
    outs[tag] = y.clone()
    print(tag, "align", lin._align_mode, "x_cap", lin._x_cap, "bf16 route", lin._uses_bf16_route(), "w_packed", lin._w_packed is not None,
          "flavour", None if lin._w_packed is None else lin._w_packed.row_scale_flavour, "pending", lin._pending_flavour is not None)
d = (outs["resident"] - outs["packed"]).abs().max().item()
print("max diff", d, "of", outs["resident"].abs().max().item())
