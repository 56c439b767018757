"""Exception statistics of the bench operands; whole-row window statistics (what-if)."""
import sys, numpy as np, torch
sys.path.insert(0, "llm-mixed-q_amd"); sys.path.insert(0, ".")
from mi355q import ops
import bench
dev = torch.device("cuda:0")
x, w, b = bench.make_inputs(torch, dev, 0)
for name, t, tok in (("x", x, True), ("w", w, False)):
    _, m, e = ops.block_fp_quantize(t, 6, 8, 127, [1, 16], tok, want_fake=False, want_packed=True, fast_zero_blocks=True)
    al = ops.bfp_align(m, e, 5, 127)
    torch.cuda.synchronize()
    print(name, "exception blocks", int(al.sparse[0]), "flag mean", float(al.rowflag.float().mean()))
    ee = e.cpu().numpy().reshape(4096, 256).astype(np.int32)
    # whole-row window of 3 exponent values: best per-row choice
    best = np.zeros(4096, np.int64)
    for r in range(4096):
        c = np.bincount(ee[r] - ee[r].min(), minlength=8)
        best[r] = max(c[i:i + 3].sum() for i in range(len(c) - 2))
    print("  whole-row window: exceptions", int(256 * 4096 - best.sum()), "rows with none", int((best == 256).sum()))
    for G in (32, 64):
        eg = ee.reshape(4096, 256 // G, G)
        tot = 0
        for r in range(0, 4096, 16):
            for g in range(256 // G):
                for rr in range(r, r + 16):
                    c = np.bincount(eg[rr, g] - eg[rr, g].min(), minlength=8)
                    tot += G - max(c[i:i + 3].sum() for i in range(len(c) - 2))
        print(f"  group of {G} blocks: exceptions", tot)
