"""Model-level parity at OPT-125m WIDTH (H 768, FFN 3072, 12 heads): random seeded weights and token ids
(no checkpoint / dataset is available offline), depth and sequence reduced so that the numpy oracle finishes in
minutes.  GPU = registry API on the HIP path (int8-MFMA Linear, HIP fake-quant matmuls); CPU = the oracle's
quantisers in numpy (tests/test_gpu_model.py::_oracle_forward).  Also times the GPU forward at T = 2048.

    python tools/model_parity.py [layers=2] [T=512] [opt|llama]     (llama: Llama-160m width, H 768, I 2048, 12 heads)
"""
import json, math, sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from mi355q.harness import (TinyOPTConfig, TinyOPTForCausalLM, TinyLlamaConfig, TinyLlamaForCausalLM, eval_lm_perplexity,
                            expand_quant_config, expand_llama_quant_config)
from test_gpu_model import _oracle_forward, _oracle_llama_forward

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
family = sys.argv[3] if len(sys.argv) > 3 else "opt"
fused = len(sys.argv) > 4 and sys.argv[4] == "fused"        # softmax stage folded into the P V product (causal)
one_pass = len(sys.argv) > 4 and sys.argv[4] == "one_pass"  # both products, mask and softmax in one kernel
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
all_knobs = len(sys.argv) > 4 and sys.argv[4] == "all_knobs"   # one-pass attention (token-major), grouped projections,
if all_knobs:                                                   # activation and norms inside the x quantisers; 2nd forward
    W6A6_model = dict(W6A6, mi355q_fused_attention=True, mi355q_token_major_output=True, mi355q_grouped_linear=True,
                      mi355q_fused_activation=True, mi355q_fused_norm=True)
elif one_pass:
    W6A6_model = dict(W6A6, mi355q_fused_attention=True)
elif fused:
    W6A6_model = dict(W6A6, mi355q_fused_softmax=True)
else:
    W6A6_model = W6A6
torch.manual_seed(0)
if family == "llama":
    cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=layers, num_heads=12, max_positions=2048)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6_model, cfg.num_layers))
    _oracle_forward = _oracle_llama_forward
else:
    cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=layers, num_heads=12, max_positions=2048)
    model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6_model, cfg.num_layers))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.ndim == 2 and "embed" not in n:
            p.mul_(2.0)
ids = torch.randint(0, cfg.vocab_size, (1, T))
t0 = time.time()
ref = _oracle_forward(model, W6A6, ids.numpy())
t_cpu = time.time() - t0
dev = torch.device("cuda:0")
model = model.to(dev)
with torch.no_grad():
    loss = float(model(ids.to(dev), labels=ids.to(dev))[1])
    if all_knobs:     # (the first PTQ forward packs the weights; the fused paths act from the second on)
        first_loss, loss = loss, float(model(ids.to(dev), labels=ids.to(dev))[1])
modes = sorted({m._align_mode + ({120: "", -1: "+blockwise"}.get(getattr(m, "_x_cap", 120), "+post-pass") if m._align_mode == "rows" else "")
                for m in model.modules() if hasattr(m, "_align_mode") and m._align_mode})
out = {"shape": f"{'Llama-160m' if family == 'llama' else 'OPT-125m'} width, {layers} layers, T={T}" + (", softmax stage folded" if fused else ", one-pass attention" if one_pass else ", every knob on (second forward)" if all_knobs else ""), "gpu_loss": loss, "oracle_loss": ref, "abs_diff": abs(loss - ref),
       "ppl_gpu": round(math.exp(loss), 3), "ppl_oracle": round(math.exp(ref), 3), "oracle_seconds": round(t_cpu, 1),
       "linear_align_modes": modes}
# GPU timing at the perplexity-run shape (B=1, T=2048)
ids2 = torch.randint(0, cfg.vocab_size, (1, 2048)).to(dev)
with torch.no_grad():
    for _ in range(2):
        model(ids2, labels=ids2)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        model(ids2, labels=ids2)
    torch.cuda.synchronize()
out["gpu_ms_per_forward_T2048"] = round((time.time() - t0) / 5 * 1e3, 2)
if all_knobs:
    out["gpu_loss_first_forward"] = first_loss
print(json.dumps(out))
