"""Depth parity, teacher-forced: a 12-layer model at OPT-125m / Llama-160m width, T = 2048 tokens, run once on the GPU with a hook
on every quantised Linear; every Linear's output is then checked against the oracle's steady-state PTQ Linear (float64
contraction of the oracle-quantised operands, oracle/np_oracle.py linear_ptq) ON THE VERY INPUT THE MODULE SAW and the
un-quantised weights.  End-to-end losses of two implementations drift apart at depth because a last-bit difference in one
Linear's output moves the W6 rounding of a few of the next quantiser's 10^6 inputs -- this tool shows that
no Linear of the run is off by more than the summation-order noise, whatever the end-to-end loss difference is.

    python tools/teacher_forced_parity.py [layers=12] [T=2048] [opt|llama]   -> one JSON line
"""
import json, sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch
from mi355q.harness import (TinyOPTConfig, TinyOPTForCausalLM, TinyLlamaConfig, TinyLlamaForCausalLM, expand_quant_config,
                            expand_llama_quant_config)
from mi355q.quantize import get_quantized_cls
from oracle import np_oracle as O
from test_gpu_model import _oracle_forward, _oracle_llama_forward

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 12
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
family = sys.argv[3] if len(sys.argv) > 3 else "opt"
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
torch.manual_seed(0)
if family == "llama":
    cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=layers, num_heads=12, max_positions=2048)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
    oracle_forward = _oracle_llama_forward
else:
    cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=layers, num_heads=12, max_positions=2048)
    model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
    oracle_forward = _oracle_forward
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.ndim == 2 and "embed" not in n:
            p.mul_(2.0)
ids = torch.randint(0, cfg.vocab_size, (1, T))
w0 = {n: (m.weight.detach().clone().numpy(), None if m.bias is None else m.bias.detach().clone().numpy())
      for n, m in model.named_modules() if isinstance(m, get_quantized_cls("linear", W6A6)) and n.startswith("layers.")}
t0 = time.time()
ref_loss = oracle_forward(model, W6A6, ids.numpy())
t_oracle = time.time() - t0
dev = torch.device("cuda:0")
model = model.to(dev)
io = {}
for n, m in model.named_modules():
    if n in w0:
        m.register_forward_hook(lambda mod, i, o, n=n: io.__setitem__(n, (i[0].detach().cpu().numpy(), o.detach().cpu().numpy())))
with torch.no_grad():
    loss = float(model(ids.to(dev), labels=ids.to(dev))[1])
worst, worst_name, per_layer = 0.0, None, {}
for n, (xin, yout) in io.items():
    w, b = w0[n]
    want = O.linear_ptq(xin.reshape(-1, xin.shape[-1]), w, b, W6A6)[0]
    e = float(np.abs(yout.reshape(want.shape) - want).max() / np.abs(want).max())
    li = int(n.split(".")[1])
    per_layer[li] = max(per_layer.get(li, 0.0), e)
    if e > worst:
        worst, worst_name = e, n
print(json.dumps({"shape": f"{'Llama-160m' if family == 'llama' else 'OPT-125m'} width, {layers} layers, T={T}, W6A6",
                  "gpu_loss": loss, "oracle_loss": ref_loss, "end_to_end_abs_diff": abs(loss - ref_loss),
                  "linears_checked": len(io), "worst_teacher_forced_rel_err": worst, "worst_linear": worst_name,
                  "worst_rel_err_per_layer": [per_layer[i] for i in sorted(per_layer)], "oracle_seconds": round(t_oracle, 1)}))
