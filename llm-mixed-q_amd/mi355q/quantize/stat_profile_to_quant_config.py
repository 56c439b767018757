"""Statistic profile -> fixed-point (`integer`) quant config: what `cli/transform_stat_profile_to_int_config.py` and the
conditional search call after a profiling pass (reference `quantize/stat_profile_to_quant_config.py:4-78`).

A stat profile maps "<root>:<layer>:...:<entry>" (entry = data_in / weight / bias ...) to recorded ranges; for every entry
the widest fractional width is chosen under which the recorded half range still fits `width` signed bits, and the result is
nested by layer name with one node config per layer."""
from __future__ import annotations

import math


def find_int_frac_width(width: int, max_half_range: float, frac_choices=None) -> int:
    """largest frac_width with max_half_range * 2^frac_width <= 2^(width - 1) - 1 (optionally the largest allowed choice
    not above it)"""
    assert max_half_range > 0, f"max_half_range must be positive, got {max_half_range}"
    assert width > 0, f"width must be positive, got {width}"
    frac = math.floor(math.log2((2 ** (width - 1) - 1) / max_half_range))
    if frac_choices is None:
        return frac
    return max(c for c in frac_choices if c <= frac)


def create_nested_dict(d: dict, key_list: list, value) -> None:
    """d[k0][k1]...[kn] = value, merging into a dict already at the leaf; a non-dict leaf in the way is an error"""
    *path, leaf = key_list
    for key in path:
        d = d.setdefault(key, {})
    if leaf not in d:
        d[leaf] = value
    elif isinstance(d[leaf], dict):
        d[leaf].update(value)
    else:
        raise ValueError(f"Cannot create nested dict at {key_list} with value {value}")


def _pick(option, name, what):
    if isinstance(option, dict):
        return option[name]
    return option


def transform_stat_profile_to_int_quant_config(stat_profile: dict, range_entry, width, frac_choices=None,
                                               root_name: str = "root", is_ptq: bool = True, bypass: bool = False) -> dict:
    if not isinstance(width, (int, dict)):
        raise ValueError(f"Unknown type of width: {type(width)}")
    if frac_choices is not None and not isinstance(frac_choices, (dict, list, tuple)):
        raise ValueError(f"Unknown type of frac_choices: {type(frac_choices)}")
    quant_config: dict = {}
    for full_name, stat in stat_profile.items():
        rng = stat[range_entry]
        half_range = max(abs(rng["min"]), abs(rng["max"]))
        entry_width = width[f"{full_name}_width"] if isinstance(width, dict) else width
        choices = _pick(frac_choices, full_name, "frac_choices")
        frac = find_int_frac_width(entry_width, half_range, choices)
        *layers, entry = full_name.removeprefix(f"{root_name}:").split(":")
        create_nested_dict(quant_config, layers, {"bypass": bypass, "name": "integer", "is_ptq": is_ptq,
                                                  f"{entry}_width": entry_width, f"{entry}_frac_width": frac})
    return quant_config
