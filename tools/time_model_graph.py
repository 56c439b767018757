"""Eager vs HIP-graph replay of the 12-layer OPT-125m-width / Llama-160m-width forward at T = 2048 (mi355q/graphs.py)."""
import json, sys, time
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
from mi355q.graphs import GraphedForward
from mi355q.harness import (TinyOPTConfig, TinyOPTForCausalLM, TinyLlamaConfig, TinyLlamaForCausalLM, expand_quant_config,
                            expand_llama_quant_config)
W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
            mi355q_fused_softmax=True, mi355q_fused_attention=(len(sys.argv) > 1 and sys.argv[1] == "one_pass"))
dev = torch.device("cuda:0")
for family in ("opt", "llama"):
    torch.manual_seed(0)
    if family == "opt":
        cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=12, num_heads=12, max_positions=2048)
        model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
    else:
        cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=12, num_heads=12, max_positions=2048)
        model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
    model = model.to(dev).eval()
    ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to(dev)

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t0) / n * 1e3

    with torch.no_grad():
        ref = model(ids)[0].clone()
        t_eager = timed(lambda: model(ids))
    fwd = GraphedForward(lambda t: model(t)[0], (ids,))
    same = bool(torch.equal(fwd(ids), ref))
    t_graph = timed(lambda: fwd(ids))
    print(json.dumps({"model": f"{'OPT-125m' if family == 'opt' else 'Llama-160m'} width, 12 layers, T=2048, W6A6, " + ("one-pass attention" if W6A6["mi355q_fused_attention"] else "softmax folded into P V"),
                      "eager_ms": round(t_eager, 3), "graph_ms": round(t_graph, 3), "graph_equals_eager": same}), flush=True)
