"""In-kernel phase stamps of the one-pass attention kernel (library built with -DATTN_STAMPS: `make -C llm-mixed-q_amd/csrc EXTRA=-DATTN_STAMPS`
after touching mi355q_attention.hip): medians over the workgroups, us.   python tools/dbg/attn_stamps.py [heads T D]"""
import os, sys, ctypes
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import mi355q.quantize as Q
from mi355q import _lib
dev = torch.device('cuda:0')
H, T, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 2048, 128)
cfg = dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16])
g = torch.Generator().manual_seed(1)
q, k, v = (torch.randn(H, T, D, generator=g).to(dev) for _ in range(3))
fn = Q.get_quantized_func("attention", cfg)
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libmi355q.so"))
nwg = H * ((T + 31) // 32) * 2
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
for _ in range(10): fn(q, k, v, cfg, cfg, causal=True, scale_div=D ** 0.5)
torch.cuda.synchronize()
lib.mi355q_debug_attn_stamps(ctypes.c_void_p(stamps.data_ptr()))
names = ["tables + Q", "scores (K loads, MFMA, mask, max)", "max exchange", "exp + sum", "sum exchange", "quotient, quantise, P V", "reduce + store"]
for rep in range(2):
    stamps.zero_()
    fn(q, k, v, cfg, cfg, causal=True, scale_div=D ** 0.5)
    torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    d = np.diff(s, axis=1) * 0.01
    whole = (s[:, 7] - s[:, 0]) * 0.01
    order = np.argsort(whole)
    print(f"{len(s)} workgroups; whole: median {np.median(whole):.1f} max {whole.max():.1f} us; kernel span {(s[:, 7].max() - s[:, 0].min()) * 0.01:.1f} us")
    for sel, label in ((order[-len(s) // 8:], "heaviest eighth"), (order[: len(s) // 8], "lightest eighth"), (order, "all")):
        print(f"  {label:16s}: " + "  ".join(f"{n} {np.median(d[sel, i]):.2f}" for i, n in enumerate(names)))
