"""Per-layer expansion of a model-level quantisation config.

Same contract as the reference's `models/opt_quantized/quant_config_opt.py:36-113` and
`models/llama_quantized/quant_config_llama.py:40-130`: a TOML-level dict with a mandatory `[default]`
section, optional op-level sections (`[linear]`, `[bmm]` / `[matmul]`, `[rotary_positional_encoding]`), an
optional `[model_layer]` section for every layer and `[model_layer_<i>]` sections for single layers
(mixed-precision search results are saved in that form, `search/search.py`), expanded into
`{"model_layer_<i>": {node: parse_node_config(...)}, "default": ...}`.  "NA" stands for None
(`utils/config_load.py:6-22`).  One table per model family instead of the reference's spelled-out dicts."""
from __future__ import annotations

from copy import deepcopy

from .quant_config_parser import parse_node_config

# family -> (op-level sections: name -> mase op, layer layout: group (None = top level) -> node -> op-level section)
_FAMILIES = {
    "opt": ({"linear": "linear", "bmm": "matmul"},
            {"self_attn": {"q_proj": "linear", "k_proj": "linear", "v_proj": "linear", "out_proj": "linear",
                           "bmm_0": "bmm", "bmm_1": "bmm"},
             None: {"fc1": "linear", "fc2": "linear"}}),
    "llama": ({"linear": "linear", "rotary_positional_encoding": "rotary_positional_encoding", "matmul": "matmul"},
              {"self_attn": {"q_proj": "linear", "k_proj": "linear", "v_proj": "linear", "o_proj": "linear",
                             "rotary_positional_encoding": "rotary_positional_encoding",
                             "matmul_0": "matmul", "matmul_1": "matmul"},
               "mlp": {"gate_proj": "linear", "down_proj": "linear", "up_proj": "linear"}}),
}


def convert_str_na_to_none(d):
    if isinstance(d, dict):
        return {k: convert_str_na_to_none(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return type(d)(convert_str_na_to_none(v) for v in d)
    return None if isinstance(d, str) and d == "NA" else d


def _load(config):
    assert isinstance(config, (str, dict, type(None))), "Must provide either a path, None or a dict"
    if isinstance(config, str):
        import tomli
        with open(config, "rb") as f:
            config = tomli.load(f)
    return convert_str_na_to_none(config)


def _expand(family: str, config, num_hidden_layers: int, strict: bool):
    if config is None:
        return None
    config = _load(config)
    assert "default" in config, "Must provide default config for by_name_parser"
    sections, layout = _FAMILIES[family]
    default_qc = config["default"]
    op_defaults = {sec: parse_node_config(config.get(sec, default_qc), mase_op=op) for sec, op in sections.items()}
    general = config.get("model_layer", None)
    parsed = {}
    for i in range(num_hidden_layers):
        layer_qc = config.get(f"model_layer_{i}", general) or {}
        layer = {}
        for group, nodes in layout.items():
            src = layer_qc if group is None else layer_qc.get(group, {})
            dst = layer if group is None else layer.setdefault(group, {})
            for node, sec in nodes.items():
                dst[node] = deepcopy(parse_node_config(src.get(node, op_defaults[sec]), sections[sec], strict=strict))
        parsed[f"model_layer_{i}"] = layer
    parsed["default"] = default_qc
    return parsed


def parse_opt_quantized_config(config, num_hidden_layers: int, strict: bool = True):
    return _expand("opt", config, num_hidden_layers, strict)


def parse_llama_quantized_config(config, num_hidden_layers: int, strict: bool = True):
    return _expand("llama", config, num_hidden_layers, strict)
