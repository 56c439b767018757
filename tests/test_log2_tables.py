"""The integer log2 decision tables compiled into the HIP kernels
(llm-mixed-q_amd/csrc/log2_tables.inc, from tools/gen_log2_tables.py, 80-digit decimal)
against the oracle's independent model (float64 log2 rounded once to fp32), on dense
sweeps around every decision boundary, and against torch-CPU's own log2."""
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import np_oracle as O

INC = Path(__file__).resolve().parents[1] / "llm-mixed-q_amd" / "csrc" / "log2_tables.inc"


def _tables():
    txt = INC.read_text()
    out = {}
    for name in ("ceil_thr", "floor_thr", "rnd_lo", "rnd_hi"):
        body = re.search(rf"mi355q_log2_{name}\[\d+\] = \{{(.*?)\}};", txt, re.S).group(1)
        out[name] = np.array([int(v.rstrip("u"), 16) for v in re.findall(r"0x[0-9a-f]+u", body)], dtype=np.int64)
    return out


T = _tables()
KS = np.arange(-126, 128)          # normal binades


def _vals(k, ms):
    bits = (np.uint32(k + 127) << np.uint32(23)) | ms.astype(np.uint32)
    return bits.view(np.float32)


def _table_decide(k, ms):
    i = k + 149
    ceil = k + ((ms >= T["ceil_thr"][i]) & (ms != 0))
    floor = k + (ms >= T["floor_thr"][i])
    even = k if k % 2 == 0 else k + 1
    rnd = np.where(ms < T["rnd_lo"][i], k, np.where(ms > T["rnd_hi"][i], k + 1, even))
    return ceil, floor, rnd


def _sweep(k):
    i = k + 149
    around = np.arange(-40, 41)
    ms = np.concatenate([np.arange(0, 200), (1 << 23) - 1 - np.arange(0, 200),
                         T["rnd_lo"][i] + around, T["rnd_hi"][i] + around,
                         np.random.default_rng(k + 1000).integers(0, 1 << 23, 300)])
    return np.unique(np.clip(ms, 0, (1 << 23) - 1))


@pytest.mark.parametrize("k", list(KS[::3]) + [-126, -1, 0, 1, 127])
def test_tables_agree_with_float64_model(k):
    ms = _sweep(int(k))
    lg = O.log2_f32(_vals(int(k), ms))
    ceil, floor, rnd = _table_decide(int(k), ms)
    assert np.array_equal(ceil, np.ceil(lg).astype(np.int64))
    assert np.array_equal(floor, np.floor(lg).astype(np.int64))
    assert np.array_equal(rnd, np.rint(lg).astype(np.int64))


def test_tables_cover_subnormals():
    """k in [-149,-127]: subnormal inputs, normalised by the kernels before the lookup."""
    for k in range(-149, -126):
        p = k + 149                       # position of the leading bit in the 23-bit subnormal field
        fr = np.unique(np.random.default_rng(k + 1000).integers(0, 1 << p, 64)) if p > 0 else np.array([0])
        raw = (np.uint32(1) << np.uint32(p)) | fr.astype(np.uint32)
        v = raw.view(np.float32)
        ms = (fr.astype(np.int64) << (23 - p))
        lg = O.log2_f32(v)
        ceil, floor, rnd = _table_decide(k, ms)
        assert np.array_equal(ceil, np.ceil(lg).astype(np.int64))
        assert np.array_equal(floor, np.floor(lg).astype(np.int64))
        assert np.array_equal(rnd, np.rint(lg).astype(np.int64))


def test_tables_agree_with_torch_cpu_log2():
    """The reference's actual primitive.  torch-CPU log2 is not correctly rounded everywhere
    (about 6e-5 of random inputs differ in the last bit) but every decision taken from it at
    the boundaries matches the tables."""
    torch = pytest.importorskip("torch")
    for k in list(range(-126, 128, 5)) + [-1, 0, 1]:
        ms = _sweep(k)
        lg = torch.log2(torch.from_numpy(_vals(k, ms).copy())).numpy()
        ceil, floor, rnd = _table_decide(k, ms)
        assert np.array_equal(ceil, np.ceil(lg).astype(np.int64)), k
        assert np.array_equal(floor, np.floor(lg).astype(np.int64)), k
        assert np.array_equal(rnd, np.rint(lg).astype(np.int64)), k
