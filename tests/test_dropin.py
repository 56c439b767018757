"""The registry as the reference's model classes see it (tests/golden/dropin.json, written by tools/check_dropin.py from the
REFERENCE's OPTQuantizedForCausalLM / LlamaQuantizedForCausalLM: modeling_opt.py:174-177, modeling_llama.py:208-210,289-344):
every quantised Linear those models build has, in our registry, the same class name, the same repr, the same state-dict keys;
the per-layer config expansion gives the same parsed dict; the package's host utilities give the same outputs."""
import json
import re
from pathlib import Path

import pytest

FIX = json.loads((Path(__file__).parent / "golden" / "dropin.json").read_text())
CASES = [k for k in FIX if k not in ("host_utils", "_what")]


def _node_config(parsed, name):
    """the node config of module `name` in the reference's parsed config: model.decoder.layers.0.self_attn.q_proj ->
    parsed["model_layer_0"]["self_attn"]["q_proj"]; Llama's mlp projections sit under "mlp" """
    m = re.search(r"layers\.(\d+)\.(.*)$", name)
    node = parsed[f"model_layer_{m.group(1)}"]
    for part in m.group(2).split("."):
        node = node[part]
    return node


@pytest.mark.parametrize("case", CASES)
def test_quantised_linears_look_like_the_reference_s(case):
    import torch  # noqa: F401
    import mi355q.quantize as Q
    rec = FIX[case]
    shapes = dict((k, v) for k, v in rec["state_dict"])
    assert rec["linear_reprs"], "the fixture holds no quantised Linear"
    for name, ref_repr in rec["linear_reprs"].items():
        if "layers." not in name:
            continue                                     # (lm_head / score: plain nn.Linear in the reference too)
        cfg = _node_config(rec["parsed_quant_config"], name)
        cls = Q.get_quantized_cls("linear", cfg)
        out_f, in_f = shapes[name + ".weight"]
        lin = cls(in_f, out_f, bias=(name + ".bias") in shapes, config=cfg)
        assert type(lin).__name__ == ref_repr.split("(")[0], name
        assert repr(lin) == ref_repr, name
        assert sorted(lin.state_dict().keys()) == rec["linear_state_keys"][name], name


@pytest.mark.parametrize("case", CASES)
def test_config_expansion_matches_the_reference_s(case):
    from mi355q.quantize.model_quant_config import parse_llama_quantized_config, parse_opt_quantized_config
    rec = FIX[case]
    parsed = rec["parsed_quant_config"]
    layers = len([k for k in parsed if k.startswith("model_layer_")])
    src = {"default": parsed["default"]}
    if case.startswith("llama") and parsed.get("rotary_positional_encoding") != parsed["default"]:
        src["rotary_positional_encoding"] = parsed["model_layer_0"]["self_attn"]["rotary_positional_encoding"]
    ours = (parse_llama_quantized_config if case.startswith("llama") else parse_opt_quantized_config)(json.loads(json.dumps(src)), layers)
    assert json.loads(json.dumps(ours)) == parsed


def test_package_exports_what_the_reference_s_callers_import():
    """models/*/sampler_*.py, quant_config_*.py, profiler_*.py, cli/transform_stat_profile_to_int_config.py, search/*.py"""
    import mi355q.quantize as Q
    for name in ("parse_node_config", "sample_a_dict_of_list", "QUANTIZED_FUNC_MAP", "QUANTIZED_MODULE_MAP", "QUANTIZER_MAP",
                 "profile_linear_layer", "profile_matmul_layer", "update_profile", "transform_stat_profile_to_int_quant_config",
                 "get_quantized_cls", "get_quantized_func", "get_quantizer"):
        assert hasattr(Q, name), name


def test_host_utilities_match_the_reference_s_outputs():
    import sys
    sys.path.insert(0, str(Path(__file__).parents[1] / "tools"))
    import mi355q.quantize as Q
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_dropin", Path(__file__).parents[1] / "tools" / "check_dropin.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.host_utils(Q.sample_a_dict_of_list, Q.transform_stat_profile_to_int_quant_config) == FIX["host_utils"]
    # error behaviour of the reference's (quant_config_sampler.py:12,22; stat_profile_to_quant_config.py:5-6,20-23)
    with pytest.raises(AssertionError):
        Q.sample_a_dict_of_list(mod._Trial(), "x", ["not", "a", "dict"])
    from mi355q.quantize.stat_profile_to_quant_config import create_nested_dict, find_int_frac_width
    with pytest.raises(AssertionError):
        find_int_frac_width(8, 0.0)
    d = {"a": 1}
    with pytest.raises(ValueError):
        create_nested_dict(d, ["a"], {"x": 2})
