"""Drop-in for the reference's `llm_mixed_q.models.quantize` package (its `__init__.py:1-21`):
the three registries and their getters, the config parser, the search-space sampler, the layer profiler and the
stat-profile transform -- every name the reference's model sub-packages, CLIs and search import from it
(tools/check_dropin.py builds the reference's own model classes on top of this package)."""
from .quant_config_parser import parse_node_config
from .quant_config_sampler import sample_a_dict_of_list
from .stat_profile_to_quant_config import transform_stat_profile_to_int_quant_config
from .quantized_functions import QUANTIZED_FUNC_MAP
from .quantized_layer_profiler import (profile_linear_layer, profile_matmul_layer, register_a_stat_hook,
                                       update_profile)
from .quantized_modules import QUANTIZED_MODULE_MAP, fp32_linear, gated_mlp, grouped_linear, relu_mlp      # grouped_linear: an addition (q / k / v in one launch)
from .quantizers import QUANTIZER_MAP


def get_quantized_cls(op: str, config: dict):
    return QUANTIZED_MODULE_MAP[op][config["name"]]


def get_quantized_func(op: str, config: dict):
    return QUANTIZED_FUNC_MAP[op][config["name"]]


def get_quantizer(op: str, config: dict):
    return QUANTIZER_MAP[config["name"]]
