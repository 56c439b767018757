"""Pin oracle/torch_port.py (the cpu_baseline workload) against the golden vectors."""
import json

import numpy as np
import pytest

from tests.conftest import GOLDEN

_META = json.loads((GOLDEN / "quantizers.json").read_text())
_BFP = sorted(t for t, m in _META.items() if m["quantizer"] == "block_fp")


@pytest.mark.parametrize("tag", _BFP)
def test_torch_port_block_fp(tag, golden_quantizers):
    torch = pytest.importorskip("torch")
    from oracle import torch_port as P
    meta, data = golden_quantizers
    m = meta[tag]
    y = P.block_fp_quantize(torch.from_numpy(data[f"{tag}/x"].copy()), skip_first_dim=m["skip_first_dim"], **m["params"])
    assert np.array_equal(y.numpy(), data[f"{tag}/y"], equal_nan=True)


_BM = sorted(t for t, m in _META.items() if m["quantizer"] == "block_minifloat")
_BL = sorted(t for t, m in _META.items() if m["quantizer"] == "block_log")


@pytest.mark.parametrize("tag", _BM + _BL)
def test_torch_port_block_minifloat_and_block_log(tag, golden_quantizers):
    """the cpu_baseline legs of bench.py's config-5 summary: the torch-op-order ports against the reference's outputs"""
    torch = pytest.importorskip("torch")
    from oracle import torch_port as P
    meta, data = golden_quantizers
    m = meta[tag]
    fn = P.block_minifloat_quantize if m["quantizer"] == "block_minifloat" else P.block_log_quantize
    y = fn(torch.from_numpy(data[f"{tag}/x"].copy()), skip_first_dim=m["skip_first_dim"], **m["params"])
    assert np.array_equal(y.numpy(), data[f"{tag}/y"], equal_nan=True)
