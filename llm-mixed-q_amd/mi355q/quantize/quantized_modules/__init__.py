from .linear import (LinearBlockFP, LinearBlockLog, LinearBlockMinifloat, LinearInteger, LinearLog,
                     LinearMinifloatDenorm, LinearMinifloatIEEE, fp32_linear, gated_mlp, grouped_linear, relu_mlp)

# same keys as the reference's quantized_modules/__init__.py:5-15
QUANTIZED_MODULE_MAP = {
    "linear": {
        "block_fp": LinearBlockFP,
        "integer": LinearInteger,
        "minifloat_ieee": LinearMinifloatIEEE,
        "minifloat_denorm": LinearMinifloatDenorm,
        "block_log": LinearBlockLog,
        "log": LinearLog,
        "block_minifloat": LinearBlockMinifloat,
    },
}
