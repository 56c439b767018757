"""Llama-160m-width forward at T=2048 (random weights) for rocprofv3 kernel stats."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q.harness import TinyLlamaConfig, TinyLlamaForCausalLM, expand_llama_quant_config
from mi355q import ops
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
torch.manual_seed(0)
cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=12, num_heads=12, max_positions=2048)
model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(dict(W6A6, mi355q_fused_attention=True), cfg.num_layers))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.ndim == 2 and "embed" not in n: p.mul_(1.0)
model = model.to("cuda:0")
ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to("cuda:0")
with torch.no_grad():
    for _ in range(6):
        model(ids, labels=ids)
torch.cuda.synchronize()
for n, m in model.named_modules():
    if hasattr(m, "_align_mode") and n.startswith("layers.0."):
        wa = m._packed[0]
        fill = ops.row_list_fill(wa.sparse, wa.rows) if wa.row_aligned else int(wa.sparse[0])
        print(n, m._align_mode, getattr(m, "_x_cap", None), "w fill", fill, file=sys.stderr)
