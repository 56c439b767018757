"""Analytic per-layer statistics ("model-stat API") with the reference's definitions
(`models/quantize/quantized_layer_profiler.py:18-27, 30-119, 122-177, 180-206`): parameter and
activation counts, storage bits per format, FLOPs = M*N*(2K-1) (+M*N for bias)."""
from __future__ import annotations

import numpy as np


def compute_tensor_bits_fp(tensor_shape: np.ndarray, width: int):
    return np.prod(tensor_shape) * width


def compute_tensor_bits_integer(tensor_shape: np.ndarray, width: int):
    return np.prod(tensor_shape) * width


def compute_tensor_bits_block_fp(tensor_shape: np.ndarray, width: int, exponent_width: int, block_size: np.ndarray):
    """elements (padded to whole blocks) * width + one shared exponent per block"""
    if tensor_shape.size > block_size.size:
        block_size = np.append([1] * (tensor_shape.size - block_size.size), block_size)
    elif tensor_shape.size < block_size.size:
        block_size = block_size[-tensor_shape.ndim:]
    num_blocks = np.prod(np.ceil(tensor_shape / block_size))
    return num_blocks * np.prod(block_size) * width + num_blocks * exponent_width


def _tensor_bits(cfg: dict, prefix: str, shape: np.ndarray):
    arith = cfg["name"]
    if cfg.get("bypass", False):
        return compute_tensor_bits_fp(shape, 32)
    if arith == "integer":
        return compute_tensor_bits_integer(shape, cfg[f"{prefix}_width"])
    if arith == "block_fp":
        return compute_tensor_bits_block_fp(shape, cfg[f"{prefix}_width"], cfg[f"{prefix}_exponent_width"],
                                            np.array(cfg[f"{prefix}_block_size"]))
    raise ValueError(f"Unknown quant_arith: {arith}")


def _as_profile(num_params, num_acts, param_bits, act_bits, flops) -> dict:
    r = lambda v: np.rint(v).astype(np.int64)
    return {"num_params": r(num_params), "num_acts": r(num_acts), "param_bits": r(param_bits),
            "act_bits": r(act_bits), "flops": r(flops)}


def profile_linear_layer(quant_config: dict, in_features: int, out_features: int, bias: bool, batch_size: int):
    w_shape, b_shape = np.array((in_features, out_features)), np.array((out_features,))
    x_shape = np.array((batch_size, in_features))
    # the reference reads these unconditionally (KeyError if absent), even when bypassed
    quant_config["weight_width"], quant_config["data_in_width"]
    if bias:
        quant_config["bias_width"]
    p_bits = _tensor_bits(quant_config, "weight", w_shape)
    if bias:
        p_bits = p_bits + _tensor_bits(quant_config, "bias", b_shape)
    x_bits = _tensor_bits(quant_config, "data_in", x_shape)
    flops = batch_size * out_features * (2 * in_features - 1) + (batch_size * out_features if bias else 0)
    return _as_profile(in_features * out_features + (out_features if bias else 0), batch_size * in_features,
                       p_bits, x_bits, flops)


def profile_matmul_layer(quant_config: dict, data_in_0_size, data_in_1_size):
    x0_shape, x1_shape = np.array((data_in_0_size,)), np.array((data_in_1_size,))
    quant_config["data_in_width"]
    if quant_config.get("bypass", False) or quant_config["name"] == "integer":
        x_bits = _tensor_bits(quant_config, "data_in", x0_shape) + _tensor_bits(quant_config, "data_in", x1_shape)
    elif quant_config["name"] == "block_fp":
        # operand 1 uses the data_in width with the weight exponent width / block size (reference :160-170)
        x_bits = _tensor_bits(quant_config, "data_in", x0_shape) + compute_tensor_bits_block_fp(
            x1_shape, quant_config["data_in_width"], quant_config["weight_exponent_width"],
            np.array(quant_config["weight_block_size"]))
    else:
        raise ValueError(f"Unknown quant_arith: {quant_config['name']}")
    flops = data_in_0_size[0] * data_in_1_size[1] * (2 * data_in_0_size[1] - 1)
    return _as_profile(0, np.prod(x0_shape) + np.prod(x1_shape), 0, x_bits, flops)


def update_profile(profile, delta):
    for k in ("num_params", "num_acts", "param_bits", "act_bits", "flops"):
        profile[k] += delta[k]
    return profile


def register_a_stat_hook(stat_manager, name: str, module, entry: str):
    """forward(-pre) hook registration the statistic profiler uses; needs the quantised Linear to
    stay an nn.Module with real .weight/.bias Parameters (it does)."""
    if entry == "data_in":
        module.register_forward_pre_hook(stat_manager.get_pre_forward_act_hook(name))
    elif entry == "weight":
        module.register_forward_pre_hook(stat_manager.get_pre_forward_weight_hook(name, weight_name="weight"))
    elif entry == "bias":
        module.register_forward_pre_hook(stat_manager.get_pre_forward_weight_hook(name, weight_name="bias"))
    elif entry == "data_out":
        module.register_forward_hook(stat_manager.get_post_forward_act_hook(name))
    else:
        raise ValueError(f"Unknown entry: {entry}")
