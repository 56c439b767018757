"""Which route every Linear of a 2-layer BASELINE-width model takes under `auto`, with the exception-bucket fills behind the choice."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
from mi355q.harness import (TinyOPTConfig, TinyOPTForCausalLM, TinyLlamaConfig, TinyLlamaForCausalLM, expand_quant_config,
                            expand_llama_quant_config)
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
            mi355q_fused_attention=True)
if len(sys.argv) > 2:                       # width: e.g. 4 for W4A4
    wd = int(sys.argv[2])
    W6A6.update(data_in_width=wd, weight_width=wd, bias_width=wd)
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "opt1.3b":
    cfg = TinyOPTConfig(vocab_size=2048, hidden_size=2048, ffn_dim=8192, num_layers=2, num_heads=32, max_positions=2048)
    model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
else:
    cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=4096, intermediate_size=11008, num_layers=2, num_heads=32, max_positions=2048)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
model = model.to("cuda:0").eval()
ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to("cuda:0")
seen = {}
orig = ops.row_list_fill
def spy(lst, rows, bucket_cap=None):
    r = orig(lst, rows, bucket_cap)
    seen.setdefault("fills", []).append((rows, r))
    return r
ops.row_list_fill = spy
with torch.no_grad():
    model(ids)
fills = iter(seen.get("fills", []))
for n, m in model.named_modules():
    if hasattr(m, "_align_mode") and m._align_mode:
        route = "bf16 blocks" if m._uses_bf16_route() else f"int8 {m._align_mode}"
        print(f"{n:32s} {m.in_features:6d} -> {m.out_features:6d}  {route:14s} x_cap {m._x_cap}")
print("row_list_fill calls (rows, (overflowed rows, fullest bucket)) in order:", seen.get("fills"))
