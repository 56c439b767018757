#!/usr/bin/env python3
"""Drop-in check (build container only: needs /root/reference): the reference's OWN model classes
(`models/opt_quantized/modeling_opt.py`, `models/llama_quantized/modeling_llama.py`) constructed twice --

  ref     with the reference's `llm_mixed_q.models.quantize` package,
  dropin  with `llm_mixed_q.models.quantize` (and its sub-modules) aliased to `mi355q.quantize`, i.e. the registry
          `get_quantized_cls` / `get_quantized_func`, the config parser and the layer profiler of THIS repository behind the
          reference's unchanged model code (modeling_opt.py:36,174-177,246,312; modeling_llama.py:40,208-210,289-344),

each in its own interpreter, and compared: module tree (names and class names), state-dict keys and shapes, `repr(model)`,
the parsed per-layer quant config, loading one state dict into the other, and a `bypass=True` CPU forward (there is no CPU
quantiser in mi355q: a CPU forward of quantised layers must raise, which is checked too).  The `ref` summary of the quantised
Linear layers is written to tests/golden/dropin.json, which tests/test_host_logic.py pins our registry against without the
reference.

    python tools/check_dropin.py            # runs both modes, compares, writes the fixture
"""
from __future__ import annotations

import hashlib
import json
import subprocess
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd"))


def _alias_quantize():
    """llm_mixed_q.models.quantize[.*] -> mi355q.quantize[.*] before any reference model file is imported"""
    import importlib
    import mi355q.quantize as Q
    sys.modules["llm_mixed_q.models.quantize"] = Q
    for sub in ("quant_config_parser", "quantized_layer_profiler", "quantized_modules", "quantized_functions", "quantizers"):
        sys.modules[f"llm_mixed_q.models.quantize.{sub}"] = importlib.import_module(f"mi355q.quantize.{sub}")


def summary(mode: str) -> dict:
    import torch
    import gen_golden_models as G
    if mode == "dropin":
        G._stub_third_party()
        for name, path in (("llm_mixed_q", G.SRC), ("llm_mixed_q.models", G.SRC / "models")):
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m
        _alias_quantize()
    opt, optc, llama, llamac = G.load_reference_models()
    qmod = sys.modules["llm_mixed_q.models.quantize"]
    out = {"quantize_module": qmod.__name__}
    d = G.bfp_default(6, 6)
    byp = dict(d, bypass=True)
    # (the block_fp rotary function quantises whatever `bypass` says -- the reference's own quirk, kept -- so the bypassed
    #  Llama model takes an arithmetic whose rotary function honours it)
    rot = dict(name="integer", bypass=True, data_in_width=8, data_in_frac_width=7)
    for family in ("opt", "llama"):
        for tag, qcfg in (("quantised", {"default": d}), ("bypass", {"default": byp, "rotary_positional_encoding": rot} if family == "llama" else {"default": byp})):
            torch.manual_seed(0)
            if family == "opt":
                cfg = optc.OPTQuantizedConfig(vocab_size=96, hidden_size=64, num_hidden_layers=2, ffn_dim=128, max_position_embeddings=32,
                                              num_attention_heads=4, dropout=0.0, word_embed_proj_dim=64,
                                              quant_config=json.loads(json.dumps(qcfg)))
                model = opt.OPTQuantizedForCausalLM(cfg).eval()
            else:
                cfg = llamac.LlamaQuantizedConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                                  num_attention_heads=4, max_position_embeddings=32,
                                                  quant_config=json.loads(json.dumps(qcfg)))
                model = llama.LlamaQuantizedForCausalLM(cfg).eval()
            g = torch.Generator().manual_seed(1)
            with torch.no_grad():
                for n, p in model.named_parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.ndim == 1 else 0.08))
            sd = model.state_dict()
            # a state dict round trip through plain tensors (what from_pretrained does with a checkpoint)
            missing = model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
            ids = torch.randint(3, 96, (2, 12), generator=g)
            rec = {
                "modules": [[n, type(m).__name__] for n, m in model.named_modules()],
                "state_dict": [[k, list(v.shape)] for k, v in sd.items()],
                "repr_sha256": hashlib.sha256(repr(model).encode()).hexdigest(),
                "linear_reprs": {n: repr(m) for n, m in model.named_modules() if type(m).__name__.startswith("Linear")},
                "linear_state_keys": {n: sorted(m.state_dict().keys()) for n, m in model.named_modules() if type(m).__name__.startswith("Linear")},
                "parsed_quant_config": G._jsonable({k: v for k, v in cfg.quant_config.items()}),
                "load_state_dict": [list(missing.missing_keys), list(missing.unexpected_keys)],
            }
            with torch.no_grad():
                if tag == "bypass":
                    o = model(input_ids=ids, labels=ids)
                    rec["logits"] = [round(float(x), 6) for x in o.logits.flatten()[:64]]
                    rec["loss"] = round(float(o.loss), 6)
                elif mode == "dropin":
                    try:
                        model(input_ids=ids)
                        rec["cpu_forward"] = "ran"
                    except RuntimeError as e:
                        rec["cpu_forward"] = "raised: " + str(e)[:60]
            out[f"{family}_{tag}"] = rec
    # the two host utilities of the package the search / the stat-profile CLI call (quant_config_sampler.py:11-28,
    # stat_profile_to_quant_config.py:4-78) on fixed inputs
    qs = sys.modules["llm_mixed_q.models.quantize"]
    out["host_utils"] = host_utils(qs.sample_a_dict_of_list, qs.transform_stat_profile_to_int_quant_config)
    return out


class _Trial:
    """a deterministic stand-in for optuna.Trial: the i-th call picks choice (i * 7 + 3) % len(choices) and records its name"""
    def __init__(self):
        self.calls = []

    def suggest_categorical(self, name, choices):
        pick = choices[(len(self.calls) * 7 + 3) % len(choices)]
        self.calls.append(name)
        return pick


HOST_SPACE = {"name": ["block_fp"], "bypass": ["!ast!False"], "data_in_width": [6, 5, 4, 3], "data_in_block_size": ["!ast![1, 16]"],
              "weight_width": [5, 4, 3, 2], "weight_exponent_bias": ["!ast!None"], "bias_width": [5, 4, 3, 2]}
HOST_STATS = {"root:model_layer_0:self_attn:q_proj:data_in": {"range_min_max": {"min": -3.7, "max": 2.9}},
              "root:model_layer_0:self_attn:q_proj:weight": {"range_min_max": {"min": -0.31, "max": 0.27}},
              "root:model_layer_0:fc1:data_in": {"range_min_max": {"min": -11.0, "max": 14.5}},
              "root:model_layer_1:self_attn:bmm_0:data_in": {"range_min_max": {"min": -0.02, "max": 0.9}}}


def host_utils(sample, transform):
    t = _Trial()
    sampled = sample(t, "default", HOST_SPACE)
    return {"sampled": sampled, "trial_names": t.calls,
            "int_config_w8": transform(HOST_STATS, "range_min_max", 8),
            "int_config_choices": transform(HOST_STATS, "range_min_max", 6, frac_choices=[0, 2, 4, 6, 8], root_name="root", is_ptq=False, bypass=True)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] in ("ref", "dropin"):
        print("@@" + json.dumps(summary(sys.argv[1])))
        return
    res = {}
    for mode in ("ref", "dropin"):
        p = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("@@")]
        if p.returncode or not line:
            sys.exit(f"{mode} failed:\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}")
        res[mode] = json.loads(line[0][2:])
    ref, dr = res["ref"], res["dropin"]
    bad = []
    assert ref["quantize_module"] == "llm_mixed_q.models.quantize" and dr["quantize_module"] == "mi355q.quantize", (ref["quantize_module"], dr["quantize_module"])
    ok_h = ref["host_utils"] == dr["host_utils"]
    print(f"{'host_utils':18s} {'sampler / transform':20s} {'same' if ok_h else 'DIFFERENT'}")
    if not ok_h:
        bad.append(("host_utils", "outputs"))
    for case in [k for k in ref if k not in ("quantize_module", "host_utils")]:
        for field in ref[case]:
            a, b = ref[case][field], dr[case].get(field)
            if field == "logits":
                ok = max(abs(x - y) for x, y in zip(a, b)) <= 1e-5
            elif field == "loss":
                ok = abs(a - b) <= 1e-5
            else:
                ok = a == b
            if not ok:
                bad.append((case, field))
            print(f"{case:18s} {field:20s} {'same' if ok else 'DIFFERENT'}")
        cf = dr[case].get("cpu_forward")
        if cf is not None:
            print(f"{case:18s} {'cpu_forward (dropin)':20s} {cf}")
            if not cf.startswith("raised"):
                bad.append((case, "cpu_forward must raise: no CPU fallback"))
    if bad:
        for case, field in bad:
            if field in ref.get(case, {}):
                print("---", case, field, "\n ref   :", json.dumps(ref[case][field])[:1500], "\n dropin:", json.dumps(dr[case].get(field))[:1500])
        sys.exit(f"drop-in check FAILED: {bad}")
    fixture = {case: {k: ref[case][k] for k in ("linear_reprs", "linear_state_keys", "state_dict", "parsed_quant_config")}
               for case in ref if case not in ("quantize_module", "host_utils")}
    fixture["host_utils"] = ref["host_utils"]
    fixture["_what"] = ("written by tools/check_dropin.py from the REFERENCE's model classes on the reference's own quantize package; "
                        "the same classes on mi355q.quantize gave identical module trees, state dicts, reprs, parsed configs and "
                        "bypass-mode logits when this file was written")
    (ROOT / "tests" / "golden" / "dropin.json").write_text(json.dumps(fixture, indent=1))
    print("drop-in check passed; tests/golden/dropin.json written")


if __name__ == "__main__":
    main()
