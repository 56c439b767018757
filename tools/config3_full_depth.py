"""BASELINE config 3 at FULL DEPTH: a Llama-7B-SHAPED decoder (32 layers, hidden 4096, intermediate 11008, 32 heads x 128;
seeded random weights -- no checkpoint exists here) under W6A6 block_fp, block [1,16], B = 1, T = 2048, every knob on: ms per
forward eager and under HIP-graph replay, tokens / s, peak memory with the int8-resident and the width-bit packed weight
storage, and every Linear of layers 0, 15 and 31 teacher-forced against the oracle on sampled rows of the input it actually saw
(reference loop: eval/eval_lm.py:41-63; model: models/llama_quantized/modeling_llama.py:289-344).

    python tools/config3_full_depth.py [--layers 32] [--tokens 2048] [--steps 5] [--storage resident|packed] [--no-graph] [--no-parity]
    rocprofv3 --kernel-trace --stats ... -- python3 tools/config3_full_depth.py --steps 3 --no-graph --no-parity
Prints one JSON line.
"""
import argparse, json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "llm-mixed-q_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch


MIXED = "auto"        # (--no-mixed: mi355q_mixed = False -- layers whose rows fit no window go to the per-block route as a whole)
LM_HEAD = "split"     # (--vendor-head: the unquantised lm_head on torch's fp32 GEMM, as before round 6, instead of the split-bf16 product)
ATTN_OUT = True       # (--no-attn-out: the attention pass writes fp32 and o_proj's quantiser reads it, as before round 6)
ROTARY = True         # (--no-rotary: the rotary embedding as its own launch in front of the attention pass, as before round 6)
GATED = True          # (--no-gated: the grouped gate / up launch + the quantiser that reads silu(gate) * up, as before round 6)


def quant_config(storage: str, knobs: bool = True):
    d = dict(name="block_fp", bypass=False, is_ptq=True,
             data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
             weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
             bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    if knobs:
        d.update(mi355q_fused_attention=True, mi355q_grouped_linear=True, mi355q_fused_activation=True, mi355q_fused_norm=True,
                 mi355q_token_major_output=True, mi355q_fused_residual=True, mi355q_fused_gate_up=GATED, mi355q_mixed=MIXED, mi355q_fused_rotary=ROTARY, mi355q_fused_attention_output=ATTN_OUT)
    if storage in ("packed", "hybrid"):
        d.update(mi355q_weight_storage=storage)
    # the rotary tables of every shipped TOML: 8-bit fixed point (configs/quantization/bfp_6bit.toml)
    return {"default": d, "rotary_positional_encoding": dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)}


def build(layers, tokens, storage, vocab=32000, hidden=4096, inter=11008, heads=32, seed=0, knobs=True, spread=True):
    from mi355q import harness as H
    cfg = H.TinyLlamaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_layers=layers, num_heads=heads,
                            max_positions=tokens)
    torch.manual_seed(seed)
    with torch.device("cuda:0"):                       # (26 GB of fp32 weights: created where they live)
        model = H.TinyLlamaForCausalLM(cfg, H.expand_llama_quant_config(quant_config(storage, knobs), layers))
        with torch.no_grad():                          # rows of different magnitude, like trained weights (outlier channels)
            for p in model.parameters():
                if p.ndim == 2 and spread:
                    p.mul_(torch.exp(0.5 * torch.randn(p.shape[0], 1)))
    model.mi355q_lm_head = LM_HEAD
    return model.eval()


def run(layers=32, tokens=2048, steps=5, storage="resident", graph=True, parity=True, knobs=True, spread=True):
    from mi355q import graphs
    dev = torch.device("cuda:0")
    torch.cuda.reset_peak_memory_stats()
    model = build(layers, tokens, storage, knobs=knobs, spread=spread)
    ids = torch.randint(0, 32000, (1, tokens), generator=torch.Generator().manual_seed(1)).to(dev)
    want = sorted({0, layers // 2 - 1 if layers > 2 else 0, layers - 1})
    with torch.no_grad():
        model(ids)                                                   # packs every weight (first PTQ forward)
        if storage in ("packed", "hybrid", "released"):
            # ("released": resident operands, the fp32 copies dropped -- what every storage mode may do once the weights are packed)
            for m in model.modules():
                if hasattr(m, "release_fp32_weight") and getattr(m, "_packed", None) is not None:
                    m.release_fp32_weight()
            torch.cuda.empty_cache()
        model(ids)                                                   # settles the routes
        torch.cuda.synchronize()
        mem_after_pack = torch.cuda.memory_allocated() / 2**30
        # (each forward timed on its own, the median reported: a box now and then stalls a stream for tens of ms, which in a mean
        #  over five 30-ms forwards would be a third of the figure)
        per = []
        from mi355q import ops as _ops
        _ops.vendor_gemm_calls(reset=True)
        for _ in range(steps):
            t0 = time.perf_counter()
            logits, loss = model(ids, labels=ids)
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) * 1e3)
        ms_eager = sorted(per)[len(per) // 2]
        vendor = {k: v // steps for k, v in sorted(_ops.vendor_gemm_calls().items())}
        from mi355q.quantize.quantized_modules.linear import _LinearBase as _LB
        routes = {}
        for m in model.modules():
            if isinstance(m, _LB):
                r = ("bf16 per-block" if m._uses_bf16_route() else
                     "mixed (int8 class 0 + bf16 class 1)" if getattr(m, "_mixed", None) is not None else f"int8 {m._align_mode}")
                routes[r] = routes.get(r, 0) + 1
        out = {"config": "BASELINE config 3 at full depth: Llama-7B shape, W6A6 block_fp [1,16], seeded random weights",
               "layers": layers, "tokens": tokens, "weight_storage": storage, "knobs": knobs, "row_spread_weights": spread,
               "loss": round(float(loss), 5), "timing": "median of the per-forward times", "ms_per_forward_eager": round(ms_eager, 2),
               "tokens_per_s_eager": round(tokens / ms_eager * 1e3, 1),
               "resident_GiB_after_packing": round(mem_after_pack, 2),
               "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2), "linear_routes": routes,
               # (torch's own GEMMs an eager forward called: the unquantised lm_head -- as in the reference -- and nothing else)
               "vendor_gemm_calls_per_forward": vendor, "gated_mlp": GATED, "rotary_on_load": ROTARY, "attention_writes_o_proj_operand": ATTN_OUT,
               "lm_head": "fp32-equivalent split-bf16 product (mi355q_fp32_split_tile)" if LM_HEAD == "split" else "vendor fp32 GEMM"}
        if graph:
            g = graphs.GraphedForward(lambda t: model(t)[0], (ids,))
            for _ in range(2):
                g(ids)
            torch.cuda.synchronize()
            per = []
            for _ in range(steps):
                t0 = time.perf_counter()
                lg = g(ids)
                torch.cuda.synchronize()
                per.append((time.perf_counter() - t0) * 1e3)
            ms_graph = sorted(per)[len(per) // 2]
            lg0 = lg[0] if isinstance(lg, (tuple, list)) else lg
            out.update(ms_per_forward_graph=round(ms_graph, 2), tokens_per_s_graph=round(tokens / ms_graph * 1e3, 1),
                       graph_equals_eager=bool(torch.equal(lg0, logits)))
        if parity and storage == "resident":
            # teacher-forced: every Linear of the tapped layers on 48 sampled rows of the input it saw, against the oracle's
            # activation quantiser and a float64 contraction with the module's CURRENT weights (quantised in place by the
            # first PTQ forward: linear.py:63-71 -- the values F.linear receives).  The tapped layers run with their launches
            # un-grouped for this one forward so that every projection sees its own input through its own forward().
            from oracle import np_oracle as O
            from mi355q.quantize.quantized_modules.linear import _LinearBase
            io, hooks, saved = {}, [], []
            knob_keys = ("mi355q_grouped_linear", "mi355q_fused_norm", "mi355q_fused_activation")
            for li in want:
                layer = model.layers[li]
                dicts = [m.config for m in layer.modules() if isinstance(m, _LinearBase)] + list(layer.self_attn.qc.values())
                for dct in dicts:
                    for k in knob_keys:
                        if dct.get(k):
                            saved.append((dct, k))
                            dct[k] = False
                for name, mod in layer.named_modules():
                    if isinstance(mod, _LinearBase):
                        hooks.append(mod.register_forward_hook(
                            lambda m_, i, o, name=f"layers.{li}.{name}": io.__setitem__(name, (i[0].detach(), o.detach(), m_))))
            model(ids)
            for h in hooks:
                h.remove()
            for dct, k in saved:
                dct[k] = True
            worst, rng = 0.0, np.random.default_rng(0)
            for name, (xin, yout, mod) in io.items():
                x2 = xin.reshape(-1, xin.shape[-1])
                pick = np.sort(rng.choice(x2.shape[0], size=min(48, x2.shape[0]), replace=False))
                c = mod.config
                xq = O.block_fp_quantize(x2[pick].cpu().numpy(), c["data_in_width"], c["data_in_exponent_width"],
                                         c["data_in_exponent_bias"], [1, 16], True)
                ref = xq.astype(np.float64) @ mod.weight.detach().cpu().numpy().astype(np.float64).T
                got = yout.reshape(-1, yout.shape[-1])[pick].cpu().numpy()
                worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
            out.update(teacher_forced_layers=want, teacher_forced_linears=len(io), teacher_forced_worst_rel_err=float(f"{worst:.3e}"))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--storage", default="resident", choices=["resident", "released", "hybrid", "packed"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-knobs", action="store_true")
    ap.add_argument("--no-spread", action="store_true", help="plain N(0, 0.02) weights instead of rows of different magnitude")
    ap.add_argument("--no-mixed", action="store_true", help="mi355q_mixed = False")
    ap.add_argument("--no-gated", action="store_true", help="without the gated epilogue (mi355q_fused_gate_up = False)")
    ap.add_argument("--vendor-head", action="store_true", help="lm_head on torch's fp32 GEMM (before round 6)")
    ap.add_argument("--no-rotary", action="store_true", help="mi355q_fused_rotary = False")
    ap.add_argument("--no-attn-out", action="store_true", help="mi355q_fused_attention_output = False")
    a = ap.parse_args()
    ATTN_OUT = not a.no_attn_out
    ROTARY = not a.no_rotary
    LM_HEAD = "vendor" if a.vendor_head else "split"
    GATED = not a.no_gated
    MIXED = False if a.no_mixed else "auto"
    print(json.dumps(run(a.layers, a.tokens, a.steps, a.storage, not a.no_graph, not a.no_parity, not a.no_knobs, not a.no_spread)), flush=True)
