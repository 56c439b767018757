"""Pin the numpy oracle against golden vectors captured from the imported reference
(tools/gen_golden.py).  Bit-exact on the fake-quantised fp32 outputs and on every integer
the formats store (shared exponents / biases, mantissas, element exponents)."""
import json

import numpy as np
import pytest

from oracle import np_oracle as O
from tests.conftest import GOLDEN

_META = json.loads((GOLDEN / "quantizers.json").read_text())


def _same(a, b):
    """bit-for-bit on values (signed zeros compare equal; NaN == NaN)."""
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("tag", sorted(_META))
def test_quantizer_matches_reference(tag, golden_quantizers):
    meta, data = golden_quantizers
    m = meta[tag]
    x, y = data[f"{tag}/x"], data[f"{tag}/y"]
    p, skip = dict(m["params"]), m["skip_first_dim"]
    if m["quantizer"] == "block_fp":
        code = O.bfp_encode(x, skip_first_dim=skip, **p)
        assert _same(code.exp, data[f"{tag}/exp"]), "shared exponents differ"
        assert _same(code.mant, data[f"{tag}/mant"]), "mantissas differ"
        out = O.block_fp_quantize(x, skip_first_dim=skip, **p)
    elif m["quantizer"] == "block_minifloat":
        code = O.bm_encode(x, skip_first_dim=skip, **p)
        assert _same(code.bias, data[f"{tag}/bias"]), "shared bias differs"
        assert _same(code.sign, data[f"{tag}/sign"])
        assert _same(code.exp, data[f"{tag}/exp"]), "element exponents differ"
        assert _same(code.mant, data[f"{tag}/mant"]), "element mantissas differ"
        out = O.block_minifloat_quantize(x, skip_first_dim=skip, **p)
    else:
        code = O.bl_encode(x, skip_first_dim=skip, **p)
        assert _same(code.bias, data[f"{tag}/bias"]), "shared bias differs"
        assert _same(code.sign, data[f"{tag}/sign"])
        assert _same(code.exp, data[f"{tag}/exp"]), "element exponents differ"
        out = O.block_log_quantize(x, skip_first_dim=skip, **p)
    assert out.dtype == np.float32 and out.shape == y.shape
    assert _same(out, y), f"max |diff| = {np.nanmax(np.abs(out - y))}"


def test_docstring_known_answers():
    """minifloat.py:150-153: minifloat_ieee(8,4,bias 15) pattern 1 0111 011 = -0.00537109375.
    block_minifloat delegates to that function; a block whose shared bias is 15 must keep it."""
    kat = json.loads((GOLDEN / "kat.json").read_text())
    v = np.float32(kat["minifloat_ieee_8_4_bias15"]["value"])
    # block max in [2**15, 2**16) -> shared bias 15; the KAT value sits in the same block
    x = np.zeros((1, 16), np.float32)
    x[0, 0] = 40000.0
    x[0, 1] = v
    out = O.block_minifloat_quantize(x, 8, 4, 8, [1, 16], True)
    assert out[0, 1] == v
    code = O.bm_encode(x, 8, 4, 8, [1, 16], True)
    assert code.bias[0] == 15 and code.exp[0, 1] == 7 - 15 and code.mant[0, 1] == 3 and code.sign[0, 1] == -1


MODULE_TAGS = ["bfp_6bit", "bfp_4bit", "block_fp", "block_minifloat", "block_log"]


@pytest.mark.parametrize("tag", MODULE_TAGS)
@pytest.mark.parametrize("has_bias", [1, 0])
def test_linear_ptq_matches_reference(tag, has_bias, golden_modules):
    meta, data = golden_modules
    cfg = meta[tag]["linear_config"]
    k = f"{tag}/linear_bias{has_bias}"
    w = data[f"{k}/w"]
    b = data[f"{k}/b"] if has_bias else None
    for xi, yi in (("x1", "y1"), ("x2", "y2")):
        y, wq, bq = O.linear_ptq(data[f"{k}/{xi}"], w, b, cfg)
        assert _same(wq, data[f"{k}/wq"]), "in-place quantised weight differs"
        if has_bias:
            assert _same(bq, data[f"{k}/bq"])
        ref = data[f"{k}/{yi}"]
        # the reference GEMM is fp32 with its own summation order: compare to tolerance
        np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("tag", ["bfp_6bit", "bfp_4bit", "block_fp"])
def test_bfp_int_gemm_identity(tag, golden_modules):
    """integer block dots scaled by 2**(ex+ew) == F.linear on fake-quant operands."""
    meta, data = golden_modules
    cfg = meta[tag]["linear_config"]
    k = f"{tag}/linear_bias1"
    x2 = data[f"{k}/x2"]
    y_int = O.bfp_linear_int(x2, data[f"{k}/w"], data[f"{k}/b"], cfg)
    y_fq, _, _ = O.linear_ptq(x2, data[f"{k}/w"], data[f"{k}/b"], cfg)
    ref = data[f"{k}/y2"]
    scale = np.abs(ref).max()
    # elements with |x|<=1e-8 pass through unquantised in the fake-quant path: <=1e-8*K*|w| apart
    np.testing.assert_allclose(y_int, y_fq, rtol=0, atol=1e-6 * scale)
    np.testing.assert_allclose(y_int, ref, rtol=2e-5, atol=2e-5 * scale)


@pytest.mark.parametrize("tag", MODULE_TAGS)
@pytest.mark.parametrize("op", ["bmm0", "bmm1", "mm4d", "mm2d"])
def test_matmul_matches_reference(tag, op, golden_modules):
    meta, data = golden_modules
    cfg = meta[tag]["matmul_config"]
    out = O.matmul_quantized(data[f"{tag}/{op}/x"], data[f"{tag}/{op}/y"], cfg)
    ref = data[f"{tag}/{op}/out"]
    np.testing.assert_allclose(out, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


def test_elementwise_quantizers_match_reference_fixtures():
    """minifloat_ieee / minifloat_denorm / log (minifloat.py:21-196, log.py:22-56): the oracle on the fixtures that
    tools/gen_golden.py:gen_elementwise generated by running the reference, bit for bit; plus the two known-answer values of
    the reference's docstrings"""
    import json
    from pathlib import Path
    import numpy as np
    from oracle import np_oracle as O
    g = Path(__file__).parent / "golden"
    z, cases = np.load(g / "elementwise.npz"), json.loads((g / "elementwise.json").read_text())
    fns = {"minifloat_ieee": O.minifloat_ieee_quantize, "minifloat_denorm": O.minifloat_denorm_quantize, "log": O.log_quantize}
    assert len(cases) == 70
    for tag, c in cases.items():
        got = fns[c["quantizer"]](z[f"x/{c['input']}"], **c["params"])
        assert np.array_equal(got.view(np.uint32), z[f"y/{tag}"].view(np.uint32)), tag
    kat = json.loads((g / "kat.json").read_text())
    # 1 0111 011 with bias 15: (-1) * 2^(7-15) * (1 + 3/8) resp. * (3/8); both are fixed points of their quantiser
    for name, f in (("minifloat_ieee_8_4_bias15", O.minifloat_ieee_quantize), ("minifloat_denorm_8_4_bias15", O.minifloat_denorm_quantize)):
        v = np.float32(kat[name]["value"])
        assert f(np.array([v]), 8, 4, 15)[0] == v
