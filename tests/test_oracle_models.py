"""G5: the model-level oracle (oracle/np_models.py) and the per-layer config expansion against fixtures generated
from the REFERENCE's own model classes (tools/gen_golden_models.py -> tests/golden/models.*): tiny OPT / Llama,
W6A6, W4A4, mixed per-layer widths ([model_layer_i] sections), K % 128 == 0 variants."""
import json

import numpy as np
import pytest

from tests.conftest import GOLDEN
from oracle import np_models as NM

META = json.loads((GOLDEN / "models.json").read_text())


@pytest.fixture(scope="module")
def data():
    return np.load(GOLDEN / "models.npz")


@pytest.mark.parametrize("tag", sorted(META))
def test_config_expansion_matches_reference(tag):
    from mi355q.quantize.model_quant_config import parse_llama_quantized_config, parse_opt_quantized_config
    m = META[tag]
    parse = parse_opt_quantized_config if m["family"] == "opt" else parse_llama_quantized_config
    mine = parse(json.loads(json.dumps(m["quant_config"])), m["num_layers"])
    assert mine == m["parsed_quant_config"]


def test_mixed_config_really_is_mixed():
    qc = META["opt_mixed"]["parsed_quant_config"]
    assert qc["model_layer_0"]["fc2"]["weight_width"] == 2 and qc["model_layer_1"]["fc2"]["weight_width"] == 6
    assert qc["model_layer_0"]["self_attn"]["bmm_0"]["data_in_width"] == 5
    assert qc["model_layer_0"]["fc1"]["data_in_exponent_bias"] is None          # "NA" -> None


@pytest.mark.parametrize("tag", sorted(META))
def test_oracle_model_forward_matches_reference(tag, data):
    sd, qc, ids, ref_logits, ref_loss, m = NM.load_fixture(META, data, tag)
    if m["family"] == "opt":
        logits, loss = NM.opt_forward(sd, qc, ids, m["num_heads"])
    else:
        logits, loss = NM.llama_forward(sd, qc, ids, m["num_heads"], m["rms_eps"])
    assert np.abs(logits - ref_logits).max() < 1e-5
    assert abs(loss - ref_loss) < 2e-6
    # perplexity to 3 d.p. (north_star): equal after rounding, or within half a unit of the third place where the reference's
    # own fp32 loss (spacing 5e-7 here, i.e. 8e-5 in ppl) puts the two on either side of a rounding boundary
    ppl, ref_ppl = float(np.exp(loss)), float(np.exp(ref_loss))
    assert round(ppl, 3) == round(ref_ppl, 3) or abs(ppl - ref_ppl) < 1e-4, (ppl, ref_ppl)


WIDE = json.loads((GOLDEN / "models_wide.json").read_text()) if (GOLDEN / "models_wide.json").exists() else {}


@pytest.mark.parametrize("tag", sorted(WIDE))
def test_oracle_model_forward_at_real_widths(tag):
    """2 decoder layers at OPT-1.3B / Llama-7B width, the reference's own logits (tools/gen_golden_models.py --wide; weights
    regenerated from the seeded recipe).  OPT-1.3B width: to the last bits.  Llama-7B width: the first layer's attention
    output to the last bit; behind it one of ~10^6 activations per tensor rounds to the other W6 neighbour when the Linear in
    front of it sums in another order (numpy's BLAS vs torch's), and a moved value moves roundings downstream -- the logits
    agree statistically (the same holds for the HIP path: tests/test_gpu_model.py, which adds the teacher-forced check of
    every Linear)."""
    data = np.load(GOLDEN / "models_wide.npz")
    sd, qc, ids, ref_logits, ref_loss, m = NM.load_wide_fixture(WIDE, data, tag)
    taps = {}
    if m["family"] == "opt":
        logits, loss = NM.opt_forward(sd, qc, ids, m["num_heads"], taps=taps)
    else:
        logits, loss = NM.llama_forward(sd, qc, ids, m["num_heads"], m["rms_eps"], taps=taps)
    ref_attn = data[tag + "/attn0"]
    assert np.abs(taps["attn0"][:, :, :128] - ref_attn).max() <= 1e-6 * np.abs(ref_attn).max()
    d = np.abs(logits - ref_logits)
    if m["family"] == "opt":
        assert d.max() < 1e-5 and abs(loss - ref_loss) < 2e-6
    else:
        scale = np.abs(ref_logits).max()
        assert d.mean() < 5e-3 * scale and d.max() < 0.1 * scale and abs(loss - ref_loss) < 5e-3, (d.max(), d.mean(), loss, ref_loss)


# ---- trained weights (tools/gen_trained_fixture.py): a 4-layer byte-level LM trained with the reference's own classes ----
TRAINED = json.loads((GOLDEN / "trained.json").read_text()) if (GOLDEN / "trained.json").exists() else {}


def load_trained(tag):
    """(state dict fp32 from the fp16-rounded fixture weights, chunks [16, 512], meta)"""
    data = np.load(GOLDEN / "trained.npz")
    pre = tag + "/w/"
    sd = {k[len(pre):]: data[k].astype(np.float32) for k in data.files if k.startswith(pre)}
    return sd, data["input_ids"], TRAINED[tag], data


@pytest.mark.parametrize("name", ["w6a6", "w4a4"])
@pytest.mark.parametrize("tag", sorted(TRAINED))
def test_oracle_on_trained_weights(tag, name):
    """the model-level oracle on weights that are NOT noise (train loss 1.4 / 0.7 nats per byte, held-out perplexity 29 / 12
    against 260 for a random model): the reference's per-chunk losses of the first three 512-token chunks (eval_lm.py:41-63
    evaluates chunk by chunk).  The bound is the reference's own spread -- the fixture's control: the reference against itself
    with every Linear output moved by one fp32 ulp moves a chunk loss by 5e-3 ... 5e-2.  Measured over all 16 chunks
    (float64-accumulating numpy against torch's fp32 CPU GEMM): OPT W4A4 5e-7 (every rounding agrees), OPT W6A6 6e-4, Llama
    W6A6 1.2e-3, Llama W4A4 6e-3; perplexity +3e-7 / +8e-4 / +2.3e-3 / -6.1e-3.  Trained weights do not make the W6 / W4
    rounding flips go away: a flipped mantissa moves everything downstream of it."""
    sd, chunks, m, data = load_trained(tag)
    ev = m["evals"][name]
    ref = data[f"{tag}/{name}/chunk_losses"]
    bound = max(r["max_d_chunk_loss"] for r in ev["control"]["runs"])
    for c in range(3):
        ids = chunks[c][None]
        if m["family"] == "opt":
            _, loss = NM.opt_forward(sd, ev["parsed_quant_config"], ids, m["num_heads"])
        else:
            _, loss = NM.llama_forward(sd, ev["parsed_quant_config"], ids, m["num_heads"], m["rms_eps"])
        assert abs(loss - ref[c]) <= bound, (c, loss, float(ref[c]), bound)
        if tag == "opt_trained" and name == "w4a4":
            assert abs(loss - ref[c]) < 5e-6                     # (the case in which no rounding flips: the arithmetic itself agrees)


def test_trained_fixture_is_a_language_model():
    for tag, m in TRAINED.items():
        assert m["evals"]["bypass"]["perplexity"] < 40.0, (tag, m["evals"]["bypass"]["perplexity"])      # (a random model: ~260)
        assert abs(m["evals"]["w6a6"]["perplexity"] / m["evals"]["bypass"]["perplexity"] - 1.0) < 0.01
        assert m["evals"]["w4a4"]["perplexity"] > m["evals"]["w6a6"]["perplexity"]
