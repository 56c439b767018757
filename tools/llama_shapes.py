"""Llama-7B Linear shapes through the steady-state step: correctness vs the oracle on a row sample,
chosen alignment flavour, step time."""
import sys, time, json
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import mi355q.quantize as Q
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
from oracle import np_oracle as O
dev = torch.device("cuda:0")
cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
           data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
           weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
for (M, K, N) in [(2048, 4096, 4096), (2048, 4096, 11008), (2048, 11008, 4096), (2048, 768, 3072), (2048, 8192, 2048)]:
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    fp = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad():
        fp.weight.copy_(torch.randn(N, K, generator=g) * 0.02); fp.bias.copy_(torch.randn(N, generator=g) * 0.02)
    w0, b0 = fp.weight.detach().numpy().copy(), fp.bias.detach().numpy().copy()
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(dev)
    xd = x.to(dev)
    y = lin(xd)
    for _ in range(5): lin(xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): y = lin(xd)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    rows = np.arange(0, M, 97)
    ref = O.bfp_linear_int(x.numpy()[rows], w0, b0, cfg)
    err = np.abs(y.detach().cpu().numpy()[rows] - ref).max() / np.abs(ref).max()
    print(json.dumps({"M": M, "K": K, "N": N, "align": lin._align_mode, "us_per_forward": round(us, 1),
                      "TFLOP/s": round(2.0 * M * N * K / us / 1e6, 1), "max_err_over_scale": float(f"{err:.2e}")}), flush=True)
