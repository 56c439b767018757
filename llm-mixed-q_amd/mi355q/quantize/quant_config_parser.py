"""Per-op quantisation-config schema.  Mirrors the CONTRACT of the reference's
`models/quantize/quant_config_parser.py` (QUANT_ARITH_ENTRIES :32-155, MASE_OP_TO_ENTRIES
:236-267, parse_node_config :278-304): same key names, same required/optional split, same
KeyError on a missing required key, same pass-through of bypassed configs.  The tables are
generated from the per-format parameter suffixes instead of being spelled out."""
from __future__ import annotations

from copy import deepcopy

_SUFFIXES = {
    "integer": ("width", "frac_width"),
    "minifloat_ieee": ("width", "exponent_width", "exponent_bias"),
    "minifloat_denorm": ("width", "exponent_width", "exponent_bias"),
    "log": ("width", "exponent_bias"),
    "block_fp": ("width", "exponent_width", "exponent_bias", "block_size"),
    "block_minifloat": ("width", "exponent_width", "exponent_bias_width", "block_size"),
    "block_log": ("width", "exponent_bias_width", "block_size"),
}
_OPERANDS = ("weight", "data_in", "bias", "data_out")

QUANT_ARITH_ENTRIES = {
    arith: {f"{operand}_entries": tuple(f"{operand}_{s}" for s in suffixes) for operand in _OPERANDS}
    for arith, suffixes in _SUFFIXES.items()
}

# op -> (required entry groups, optional entry groups)
MASE_OP_TO_ENTRIES = {
    "add": (("name", "data_in_entries"), ("bypass",)),
    "bmm": (("name", "data_in_entries", "weight_entries"), ("bypass",)),
    "conv1d": (("name", "is_ptq", "data_in_entries", "weight_entries"), ("bias_entries", "bypass")),
    "conv2d": (("name", "is_ptq", "data_in_entries", "weight_entries"), ("bias_entries", "bypass")),
    "matmul": (("name", "data_in_entries", "weight_entries"), ("bypass",)),
    "mul": (("name", "data_in_entries"), ("bypass",)),
    "linear": (("name", "is_ptq", "data_in_entries", "weight_entries"),
               ("bias_entries", "data_out_entries", "bypass")),
    "relu": (("name", "data_in_entries"), ("bypass",)),
    "rotary_positional_encoding": (("name", "data_in_entries"), ("bypass",)),
    "sub": (("name", "data_in_entries"), ("bypass",)),
}


def cp_multi_values(src: dict, dst: dict, src_keys: tuple, dst_keys: tuple = None, strict: bool = True):
    for s, d in zip(src_keys, src_keys if dst_keys is None else dst_keys):
        if not strict and s not in src:
            continue
        dst[d] = deepcopy(src[s])


def has_multi_keys(src: dict, keys: tuple) -> bool:
    return all(k in src for k in keys)


def _keys_of(group: str, arith: str) -> tuple:
    if group in ("name", "bypass", "is_ptq"):
        return (group,)
    return QUANT_ARITH_ENTRIES[arith][group]


def optional_entry_exists(config: dict, entry_name: str) -> bool:
    stem = entry_name.removesuffix("_entries")
    return any(k.startswith(stem) for k in config)


def parse_node_config(config: dict, mase_op: str, strict: bool = True) -> dict:
    """Keep exactly the keys `mase_op` needs from `config` (required groups: KeyError when
    absent and strict; optional groups: only when some key with that prefix is present)."""
    assert mase_op in MASE_OP_TO_ENTRIES, f"Unknown mase op: {mase_op}"
    if config.get("bypass", False):
        return config
    required, optional = MASE_OP_TO_ENTRIES[mase_op]
    arith = config["name"]
    parsed: dict = {}
    for group in required:
        cp_multi_values(config, parsed, _keys_of(group, arith), strict=strict)
    for group in optional:
        if optional_entry_exists(config, group):
            cp_multi_values(config, parsed, _keys_of(group, arith), strict=strict)
    # implementation knobs of this build ("mi355q_align", "mi355q_weight_storage", "mi355q_fused_softmax", ...) ride
    # along; the reference's configs carry none, so its parsed dicts are reproduced unchanged
    for k, v in config.items():
        if k.startswith("mi355q_"):
            parsed[k] = deepcopy(v)
    return parsed
