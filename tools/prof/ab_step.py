"""A/B of the bench step (4096^3 W6A6: fused activation quantiser + tile GEMM) in ONE process, interleaved rounds (cdna guide rule 24):
round 5's row quantiser (MI355Q_QROWS_VARIANT=0: 121 registers, four workgroups a compute unit, dword stores) against round 6's
(no pre-op code, 16-byte stores, six workgroups a compute unit).  Prints the step and the quantiser alone, median / min of 7 rounds."""
import os, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
import bench
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
x, w, b = bench.make_inputs(torch, dev, 0)
_, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
wa = ops.bfp_align_rows(wm, we, 5, 127)
bq = ops.block_fp_quantize(b, 6, 8, 127, [16], False)
y = torch.empty(4096, 4096, device=dev)


def step():
    xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
    ops.bfp_gemm_aligned(xa, wa, bq, out=y)


def quant():
    ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)


def t(fn, n=200):
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


def variant(v):
    if v is None: os.environ.pop("MI355Q_QROWS_VARIANT", None)
    else: os.environ["MI355Q_QROWS_VARIANT"] = str(v)


for _ in range(600): step()
res = {("r5", "step"): [], ("r6", "step"): [], ("r5", "quant"): [], ("r6", "quant"): []}
for rnd in range(7):
    for name, v in (("r5", 0), ("r6", None)):
        variant(v)
        for _ in range(20): step()
        res[(name, "step")].append(t(step))
        res[(name, "quant")].append(t(quant))
variant(None)
out = {}
for (name, what), vals in res.items():
    vals = sorted(vals)
    out[f"{name}_{what}_us_median"] = round(vals[len(vals) // 2], 2)
    out[f"{name}_{what}_us_min"] = round(vals[0], 2)
out["what"] = "4096^3 W6A6 bench step (quantiser + GEMM, launch to launch) and the quantiser alone; r5 = MI355Q_QROWS_VARIANT=0, r6 = default; 7 interleaved rounds of 200"
print(json.dumps(out))
if os.path.isdir(ROOT / "gpurun_out"):
    open(ROOT / "gpurun_out" / "r06_step_ab.json", "w").write(json.dumps(out) + "\n")
