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
            mi355q_fused_attention="one_pass" in sys.argv, mi355q_fused_softmax="steps" not in sys.argv,
            mi355q_grouped_linear="grouped" in sys.argv, mi355q_fused_activation="fused_act" in sys.argv,
            mi355q_token_major_output="token_major" in sys.argv, mi355q_fused_norm="fused_norm" in sys.argv)
dev = torch.device("cuda:0")
families = ("opt1.3b", "llama7b") if "real_widths" in sys.argv else ("opt", "llama")
for family in families:
    torch.manual_seed(0)
    if family == "opt1.3b":      # BASELINE config 4's width, 2 layers (per-layer time = difference to the head + embedding)
        cfg = TinyOPTConfig(vocab_size=2048, hidden_size=2048, ffn_dim=8192, num_layers=2, num_heads=32, max_positions=2048)
        model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
    elif family == "llama7b":    # BASELINE config 3's width
        cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=4096, intermediate_size=11008, num_layers=2, num_heads=32, max_positions=2048)
        model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
    elif family == "opt":
        cfg = TinyOPTConfig(vocab_size=2048, hidden_size=768, ffn_dim=3072, num_layers=12, num_heads=12, max_positions=2048)
        model = TinyOPTForCausalLM(cfg, expand_quant_config(W6A6, cfg.num_layers))
    else:
        cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=12, num_heads=12, max_positions=2048)
        model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(W6A6, cfg.num_layers))
    model = model.to(dev).eval()
    ids = torch.randint(0, cfg.vocab_size, (1, 2048)).to(dev)

    def timed(fn, n=10):
        t_end = time.time() + 0.1            # (an idle MI355X needs tens of milliseconds of work to reach its clocks)
        while time.time() < t_end:
            fn()
            torch.cuda.synchronize()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t0) / n * 1e3

    with torch.no_grad():
        model(ids)                         # (the first PTQ forward packs the weights; the fused paths act from the second on)
        ref = model(ids)[0].clone()
        t_eager = timed(lambda: model(ids))
    fwd = GraphedForward(lambda t: model(t)[0], (ids,))
    same = bool(torch.equal(fwd(ids), ref))
    t_graph = timed(lambda: fwd(ids))
    label = {"opt": "OPT-125m width, 12 layers", "llama": "Llama-160m width, 12 layers", "opt1.3b": "OPT-1.3B width, 2 layers",
             "llama7b": "Llama-7B width, 2 layers"}[family]
    print(json.dumps({"model": f"{label}, T=2048, W6A6, " + ("q/k/v and gate/up grouped, " if W6A6["mi355q_grouped_linear"] else "") + ("activation inside the x quantiser, " if W6A6["mi355q_fused_activation"] else "") + ("attention output token-major, " if W6A6["mi355q_token_major_output"] else "") + ("RMSNorm inside the x quantiser, " if W6A6["mi355q_fused_norm"] else "") + ("one-pass attention" if W6A6["mi355q_fused_attention"] else "softmax folded into P V" if W6A6["mi355q_fused_softmax"] else "attention as the reference steps it"),
                      "eager_ms": round(t_eager, 3), "graph_ms": round(t_graph, 3), "graph_equals_eager": same}), flush=True)
