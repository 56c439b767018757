"""Llama-7B's q / k / v projections at 2048 tokens (three 2048 x 4096 x 4096 products against ONE activation operand) as the grouped
launch plans and tile geometries allow: the library's default plan, one launch of all three, and the small-tile geometries forced
(MI355Q_V10 = 1 | 2 | 3: 128 x 256, 256 x 128, 128 x 128 tiles of mi355q_gemm_v10.hip).  GEMMs alone, HIP events, us for the three."""
import json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
from mi355q import ops
dev = torch.device("cuda:0")
M, N, K = 2048, 4096, 4096
g = torch.Generator().manual_seed(1)
x = (torch.randn(M, K, generator=g) * torch.exp(0.5 * torch.randn(M, 1, generator=g))).to(dev)
was = []
for i in range(3):
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    was.append(ops.bfp_align_rows(wm, we, 5, 127))
xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
outs = [torch.empty(M, N, device=dev) for _ in range(3)]


def t(fn, n=50):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(e) / n * 1e3, 1)


def plan_run(plan):
    at = 0
    for gcount in plan:
        part = was[at:at + gcount]
        if gcount == 1: ops.bfp_gemm_aligned(xa, part[0], None, out=outs[at])
        else: assert ops.bfp_gemm_aligned_multi(xa, part, None, outs=outs[at:at + gcount]) is not None
        at += gcount


z = torch.randn(4096, 4096, device=dev)
for _ in range(30): z @ z
row = {"shape": "3 x (2048 x 4096 x 4096), W6A6", "default_plan": list(ops.grouped_launch_plan(M, N, 3))}
ref = None
for name, env, plan in (("default plan", None, ops.grouped_launch_plan(M, N, 3)), ("one launch of 3", None, (3,)), ("3 launches", None, (1, 1, 1)),
                        ("geom 2 (256x128), one launch", "2", (3,)), ("geom 1 (128x256), one launch", "1", (3,)), ("geom 3 (128x128), one launch", "3", (3,)),
                        ("geom 2, (2,1)", "2", (2, 1))):
    if env: os.environ["MI355Q_V10"] = env
    else: os.environ.pop("MI355Q_V10", None)
    try:
        row[name] = t(lambda: plan_run(plan))
        torch.cuda.synchronize()
        cur = torch.stack(outs).clone()
        if ref is None: ref = cur
        else: assert torch.equal(cur, ref), name
    except AssertionError as e:
        row[name] = f"n/a ({e})"
os.environ.pop("MI355Q_V10", None)
print(json.dumps(row))
if os.path.isdir(ROOT / "gpurun_out"):
    open(ROOT / "gpurun_out" / "r06_grouped_qkv.json", "w").write(json.dumps(row) + "\n")
