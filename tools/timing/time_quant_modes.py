"""Kernel-level timing of the streaming quantisers through the C ABI (outputs preallocated, so the host adds only the
ctypes call): exact vs fast zero-block mode, with / without the packed outputs.  Run on the GPU box from the repo root."""
import json
import sys

sys.path.insert(0, "llm-mixed-q_amd")
import torch
from mi355q import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load_library()
ws = ops._workspace(dev)
st = ops._stream_ptr(dev)
P = ops._ptr


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3


out = []
for name, shp in (("act[2048,4096]", (2048, 4096)), ("act[2048,11008]", (2048, 11008)), ("w[4096,4096]", (4096, 4096))):
    x = (torch.randn(*shp, generator=torch.Generator().manual_seed(7)) * 4.0).to(dev)
    rows, cols = shp
    y = torch.empty_like(x)
    mant = torch.empty(shp, dtype=torch.int8, device=dev)
    code = torch.empty(rows * cols // 16, dtype=torch.uint8, device=dev)
    cases = {
        "bfp_fake_fast": lambda: lib.mi355q_block_fp_quantize(P(x), P(y), None, None, 1, rows, cols, 1, 16, 6, 8, 127, 1, P(ws), st),
        "bfp_fake_exact": lambda: lib.mi355q_block_fp_quantize(P(x), P(y), None, None, 1, rows, cols, 1, 16, 6, 8, 127, 0, P(ws), st),
        "bfp_fake+packed_exact": lambda: lib.mi355q_block_fp_quantize(P(x), P(y), P(mant), P(code), 1, rows, cols, 1, 16, 6, 8, 127, 0, P(ws), st),
        "bfp_packed_only_exact": lambda: lib.mi355q_block_fp_quantize(P(x), None, P(mant), P(code), 1, rows, cols, 1, 16, 6, 8, 127, 0, P(ws), st),
        "bm_fake": lambda: lib.mi355q_block_minifloat_quantize(P(x), P(y), None, 1, rows, cols, 1, 16, 8, 4, 8, 0, P(ws), st),
        "bl_fake_exact": lambda: lib.mi355q_block_log_quantize(P(x), P(y), None, 1, rows, cols, 1, 16, 8, 8, 0, P(ws), st),
        "bl_fake_fast": lambda: lib.mi355q_block_log_quantize(P(x), P(y), None, 1, rows, cols, 1, 16, 8, 8, 1, P(ws), st),
    }
    for cname, fn in cases.items():
        rc = fn()
        assert rc == 0, (cname, rc)
        us = timed(fn)
        r = {"shape": name, "case": cname, "us": round(us, 2), "GB/s(8B/elt)": round(8.0 * x.numel() / us / 1e3, 1)}
        out.append(r)
        print(json.dumps(r), flush=True)
