"""Search-space sampling of a node config: the host-side hook the reference's per-model samplers
(`models/*/sampler_*.py`) call into for the Optuna search (reference `quantize/quant_config_sampler.py:11-28`).

The search space is a dict of option lists ([search_space.quant_config_seed.default] of the search TOMLs,
experiments/emnlp/configs/search/opt_1.3b_sst2.toml:24-37); an option spelled "!ast!<literal>" stands for the Python literal
(`"!ast![1, 16]"` -> [1, 16], `"!ast!None"` -> None).  `trial` is anything with Optuna's `suggest_categorical(name, choices)`
-- optuna itself is not imported, so the registry package loads without it."""
from __future__ import annotations

import ast

_AST_TAG = "!ast!"


def sample_a_list(trial, name: str, choices: list):
    """one categorical draw named `name`; literal-tagged strings are evaluated after the draw (Optuna only stores
    str / int / float / bool / None choices)"""
    if not isinstance(choices, list):
        raise AssertionError(f"choices must be a list, got {choices}")
    picked = trial.suggest_categorical(name, list(choices))       # (a fresh list: Optuna keeps a reference to it)
    if isinstance(picked, str) and picked.startswith(_AST_TAG):
        return ast.literal_eval(picked[len(_AST_TAG):])
    return picked


def sample_a_dict_of_list(trial, name: str, config: dict) -> dict:
    """{key: one draw from config[key]}; the Optuna parameter of a key is called "<name>:<key>" """
    if not isinstance(config, dict):
        raise AssertionError(f"config must be a dict, got {config}")
    return {key: sample_a_list(trial, f"{name}:{key}", options) for key, options in config.items()}
