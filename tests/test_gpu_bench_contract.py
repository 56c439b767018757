"""bench.py prints ONE JSON line with the fields the driver reads (the contract in the task description): checked on a
short run, the gemm workload through `python -m torch.distributed.run` as the driver launches it for N > 1."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE line on stdout (everything else goes to stderr)
    return json.loads(lines[0])


def _common(d, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["steps"] == steps and d["warmup"] == warmup and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert not any(k in d["config"] for k in ("model", "global_batch", "seq_len"))
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and d["value"] > 0


def test_gemm_line_through_the_launcher():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2"])
    _common(d, 6, 2)
    assert d["metric"].startswith("quantised-GEMM TFLOP/s") and d["unit"] == "TFLOP/s" and d["dtype"] == "int8"
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["unit"] == "TFLOP/s" and d["roofline"]["peak"] == 5000.0
    assert abs(d["value"] - 2 * 4096 ** 3 / (d["ms_per_step"] * 1e-3) / 1e12) / d["value"] < 2e-3
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["verify"]["ok"] is True                      # the timed step's own output against the oracle


def test_quantizers_line():
    d = _run([sys.executable, "bench.py", "--workload", "quantizers", "--steps", "3", "--warmup", "1"])
    _common(d, 3, 1)
    assert d["unit"] == "GB/s" and d["dtype"] == "f32" and d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert len(d["cases"]) == 18 and all(c["GB/s"] > 0 and c["copy_GB/s"] > 0 for c in d["cases"])
