"""Host-side cost of one PTQ Linear forward (tiny layer: the kernels take ~10 us, the rest is Python / ctypes)."""
import sys, time, cProfile, pstats
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev = torch.device("cuda:0")
cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
           data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
           weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
lin = Q.get_quantized_cls("linear", cfg)(256, 256, config=cfg).to(dev)
x = torch.randn(64, 256, device=dev)
for _ in range(70): lin(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(400): lin(x)
torch.cuda.synchronize()
print("us per forward", (time.perf_counter() - t0) / 400 * 1e6, "mode", lin._align_mode)
pr = cProfile.Profile(); pr.enable()
for _ in range(400): lin(x)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
