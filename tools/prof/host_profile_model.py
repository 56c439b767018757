"""cProfile of the eager 12-layer forward (host side): where the Python / ctypes time of the launch path goes."""
import cProfile, pstats, sys, io
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q.harness import (TinyOPTConfig, TinyOPTForCausalLM, TinyLlamaConfig, TinyLlamaForCausalLM, expand_quant_config,
                            expand_llama_quant_config)
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
            mi355q_fused_attention=True)
family = sys.argv[1] if len(sys.argv) > 1 else "llama"
torch.manual_seed(0)
if family == "opt":
    cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=12, num_heads=12, max_positions=2048)
    model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
else:
    cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=12, num_heads=12, max_positions=2048)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
model = model.to("cuda:0").eval()
ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to("cuda:0")
with torch.no_grad():
    for _ in range(3):
        model(ids)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        model(ids)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
