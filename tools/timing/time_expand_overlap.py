"""Can the expand pass of width-bit weight storage hide behind the attention pass?  The attention kernel (Llama-7B shape) leaves 32 registers
a lane free on every SIMD and is not bandwidth-bound; the int8-mode expand kernel needs 15.  Attention (pack + kernel) on one stream, the
expands of a layer's five row-scale weights (q, k, v, gate, up: 140 M values) on another: alone, alone, together.  HIP events, us."""
import json, math, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd")); sys.path.insert(0, str(ROOT))
import torch
from mi355q import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, T, D = 32, 2048, 128
q, k, v = (torch.randn(1, T, H, D, generator=g).to(dev).transpose(1, 2) for _ in range(3))
par = (6, 8, 127, 6, 8, 127)
pws = []
for n in (4096, 4096, 4096, 11008, 11008):
    w = (torch.randn(n, 4096, generator=g) * 0.02).to(dev)
    _, wm, we = ops.block_fp_quantize(w, 6, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
    pws.append(ops.pack_row_aligned_weights(wm, we, ops.bfp_align_rows(wm, we, 5, 127), 6, 127))
    del w, wm, we
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def attention():
    return ops.bfp_attention(q, k, v, par, par, causal=True, scale_div=math.sqrt(D), token_major=True)


def expands():
    with torch.cuda.stream(side):
        for i, pw in enumerate(pws):
            pw.expand(i)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(n):
        fn()
        main.wait_stream(side)
    e.record(main)
    torch.cuda.synchronize()
    return round(a.elapsed_time(e) / n * 1e3, 1)


def both():
    side.wait_stream(main)
    expands()
    attention()


def only_expands():
    side.wait_stream(main)
    expands()


for _ in range(200):                                   # (clock ramp: the first timed loop otherwise sits on it)
    attention()
torch.cuda.synchronize()
row = {"attention_us": timed(attention), "expands_of_5_weights_us": timed(only_expands), "together_us": timed(both),
       "attention_again_us": timed(attention), "together_again_us": timed(both)}
row["hidden_fraction_of_the_expands"] = round(1 - (row["together_us"] - row["attention_us"]) / row["expands_of_5_weights_us"], 2)
print(json.dumps(row))
