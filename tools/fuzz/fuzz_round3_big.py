"""The streaming quantisers on tensors around the size thresholds of their access shapes (>= 2^25 elements: zero-block map;
>= 2^26: owned pieces instead of the grid-stride loop), random all-zero block patterns, planes checked against the oracle
(each plane quantised together with the row that carries the tensor's smallest non-zero block maximum)."""
import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import mi355q.quantize as Q
from oracle import np_oracle as O
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
g = torch.Generator(device=dev).manual_seed(seed)
KW = {"block_fp": dict(width=6, exponent_width=8, exponent_bias=None, block_size=[1, 16]),
      "block_minifloat": dict(width=8, exponent_width=4, exponent_bias_width=8, block_size=[1, 16]),
      "block_log": dict(width=8, exponent_bias_width=8, block_size=[1, 16])}
bad = 0
for planes in (9, 16, 20):
    T = 2048
    x = torch.randn(planes, T, T, generator=g, device=dev) * torch.exp(torch.randn(planes, T, 1, generator=g, device=dev) * 2)
    blocks = x.view(-1, 16)
    blocks[torch.rand(blocks.shape[0], generator=g, device=dev) < 0.3] = 0          # random all-zero blocks
    x[planes // 2, 100:900] = 0                                                   # a long run of them
    bm = x.view(-1, 16).abs().amax(1)
    row = int((bm == bm[bm > 0].min()).nonzero()[0, 0]) * 16 // T
    carrier = x.view(-1, T)[row:row + 1].cpu().numpy()
    for name in ("block_log", "block_fp", "block_log", "block_minifloat"):        # (block_log twice: a hit after a foreign fill)
        got = Q.get_quantizer("", dict(name=name))(x, **KW[name], skip_first_dim=True)
        for i in (0, planes // 2, planes - 1):
            piece = np.concatenate([x[i].cpu().numpy(), carrier], 0)[None]
            want = np.asarray(getattr(O, name + "_quantize")(piece, **KW[name], skip_first_dim=True), dtype=np.float32)[0, :-1]
            gi = got[i].cpu().numpy()
            if not np.array_equal(gi, want):                      # (values; the sign of a zero is not pinned)
                bad += 1
                d = np.argwhere(gi != want)
                print("MISMATCH", planes, name, i, len(d), d[:3].tolist())
        del got
    print("planes", planes, "elements 2^%.2f" % np.log2(x.numel()), "bad", bad, flush=True)
    del x
sys.exit(1 if bad else 0)
