"""A/B of the row quantiser's variants (round 6) in ONE process, interleaved rounds (cdna guide rule 24): MI355Q_QROWS_VARIANT
0 = the shipped kernel, 1 = 16-byte stores, 2 = the build without the pre-op code, 3 = both, 6 / 7 = 2 / 3 under
__launch_bounds__(256, 8); MI355Q_QROWS_GRID = workgroups.  Every variant's operand is compared byte for byte with variant 0's."""
import os, sys, json
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "llm-mixed-q_amd"))
from mi355q import ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
g = lambda s: torch.Generator().manual_seed(s)
x = (torch.randn(4096, 4096, generator=g(0)) * torch.exp(torch.randn(4096, 1, generator=g(1)))).to(dev)
x2 = x.clone(); x2[::7, 64:80] *= 4096.0            # (rows with exception blocks)


def setv(v, grid):
    os.environ["MI355Q_QROWS_VARIANT"] = str(v)
    if grid: os.environ["MI355Q_QROWS_GRID"] = str(grid)
    else: os.environ.pop("MI355Q_QROWS_GRID", None)


def snapshot(t):
    xa = ops.block_fp_quantize_aligned_rows(t, 6, 8, 127)
    torch.cuda.synchronize()
    return [u.clone() for u in (xa.tiled, xa.exp, xa.gscale, xa.rowflag, xa.sparse)]


cases = [(0, 0), (1, 0), (2, 0), (3, 0), (2, 1536), (3, 1536), (3, 2048), (6, 2048), (7, 2048), (0, 2048), (3, 4096), (0, 512), (3, 768)]
setv(0, 0)
base = [snapshot(x), snapshot(x2)]
for v, grid in cases:
    setv(v, grid)
    for t, b in ((x, base[0]), (x2, base[1])):
        got = snapshot(t)
        # (the exception list's entry ORDER depends on which row reserved its slots first: compare the sorted entries)
        for i, (u, w) in enumerate(zip(got[:4], b[:4])):
            assert torch.equal(u, w), (v, grid, i)
        assert int(got[4][0]) == int(b[4][0]) and sorted(got[4][8::8 + 8 * 120][:16].tolist()) == sorted(b[4][8::8 + 8 * 120][:16].tolist()), (v, grid)
print("all variants byte-identical to variant 0")
res = {c: [] for c in cases}
for c in cases:                       # warm
    setv(*c)
    for _ in range(50): ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
torch.cuda.synchronize()
for rnd in range(5):
    for c in cases:
        setv(*c)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): ops.block_fp_quantize_aligned_rows(x, 6, 8, 127)
        e.record(); torch.cuda.synchronize()
        res[c].append(a.elapsed_time(e) / 200 * 1e3)
for c in cases:
    r = sorted(res[c])
    print(f"variant {c[0]} grid {c[1] or 1024:5d}: median {r[len(r)//2]:6.2f} us  min {r[0]:6.2f}  (back to back, incl. launch gaps)")
