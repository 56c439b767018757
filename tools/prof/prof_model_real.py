"""Two decoder layers at a BASELINE width (llama7b | opt1.3b), T = 2048, one-pass attention: for rocprofv3 kernel stats."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q.harness import (TinyOPTConfig, TinyOPTForCausalLM, TinyLlamaConfig, TinyLlamaForCausalLM, expand_quant_config,
                            expand_llama_quant_config)
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
            mi355q_fused_attention=True, mi355q_grouped_linear=True, mi355q_fused_activation=True, mi355q_token_major_output=True,
            mi355q_fused_norm=True)
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "opt1.3b":
    cfg = TinyOPTConfig(vocab_size=2048, hidden_size=2048, ffn_dim=8192, num_layers=2, num_heads=32, max_positions=2048)
    model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
else:
    cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=4096, intermediate_size=11008, num_layers=2, num_heads=32, max_positions=2048)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
model = model.to("cuda:0").eval()
ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to("cuda:0")
with torch.no_grad():
    for _ in range(11):
        model(ids)
torch.cuda.synchronize()
