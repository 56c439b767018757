"""OPT-125m-shaped forward at T=2048 (random weights) for rocprofv3 kernel stats."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q.harness import TinyOPTConfig, TinyOPTForCausalLM, expand_quant_config
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
torch.manual_seed(0)
cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=12, num_heads=12, max_positions=2048)
model = TinyOPTForCausalLM(cfg, expand_quant_config(dict(W6A6, mi355q_fused_softmax=len(sys.argv) < 2 or sys.argv[1] != "unfused"), cfg.num_layers)).to("cuda:0")
ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to("cuda:0")
with torch.no_grad():
    for _ in range(6):
        model(ids, labels=ids)
torch.cuda.synchronize()
