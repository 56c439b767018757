"""Exception statistics of every Linear of the OPT-125m-shaped forward (first two layers): which alignment the auto
policy picked, how full the exception buckets / lists of the live activations are."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q import ops
from mi355q.harness import TinyOPTConfig, TinyOPTForCausalLM, expand_quant_config
from mi355q.quantize.quantized_modules.linear import LinearBlockFP
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
torch.manual_seed(0)
cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=2, num_heads=12, max_positions=2048)
model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers)).to("cuda:0")
ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to("cuda:0")

def hook(name):
    def f(mod, inp):
        x = inp[0].reshape(-1, mod.in_features)
        xr = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
        over, mx = ops.row_list_fill(xr.sparse, xr.rows)
        line = f"{name:28s} M={x.shape[0]} K={mod.in_features} N={mod.out_features} mode={mod._align_mode} x rows: overflow={over} fullest={mx}"
        if mod._packed is not None and mod._packed[0] is not None:
            wa = mod._packed[0]
            line += f" | w rows fill={ops.row_list_fill(wa.sparse, wa.rows)}"
        zeros = float((x == 0).float().mean())
        print(line, f"| zeros={zeros:.3f}")
    return f

with torch.no_grad():
    model(ids, labels=ids)
    for n, m in model.named_modules():
        if isinstance(m, LinearBlockFP):
            m.register_forward_pre_hook(hook(n))
    model(ids, labels=ids)
torch.cuda.synchronize()
