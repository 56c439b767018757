import sys
from pathlib import Path
import numpy as np, torch
R = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(R / "llm-mixed-q_amd")); sys.path.insert(0, str(R))
import mi355q.quantize as Q
from oracle import np_oracle as O
g = torch.Generator().manual_seed(3)
T = 96
causal = torch.softmax(torch.randn(2, T, T, generator=g) * 3 + torch.full((T, T), float("-inf")).triu(1), dim=-1)
big = torch.randn(4, 64, 64, generator=g) * 50.0
big[:, ::3, 16:48] = 0.0
kw = dict(width=6, exponent_width=8, exponent_bias=None, block_size=[1, 16])
q = Q.get_quantizer("", dict(name="block_fp"))
for seq in (["big"], ["causal", "big"], ["causal", "causal", "big"]):
    for n in seq:
        t = dict(causal=causal, big=big)[n]
        got = q(t.to("cuda:0"), **kw, skip_first_dim=True).cpu().numpy()
    want = np.asarray(O.block_fp_quantize(big.numpy(), **kw, skip_first_dim=True), dtype=np.float32)
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    print(seq, len(bad), bad[:5].tolist(), [(got[tuple(b)], want[tuple(b)], big.numpy()[tuple(b)]) for b in bad[:5]])
