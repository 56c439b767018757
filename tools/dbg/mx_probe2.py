import sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from mi355q import ops
from oracle import np_oracle as O
dev = torch.device("cuda:0")
r = np.random.default_rng(1)
cfg = dict(name="block_fp", data_in_width=4, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=4, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=4, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
def run(x, w):
    wq = ops.block_fp_quantize(torch.from_numpy(w).to(dev), 4, 8, 127, [1, 16], False)
    wop = ops.block_fp_quantize_mx(wq.contiguous(), 4, 8, 127, reuse=False)
    xop = ops.block_fp_quantize_mx(torch.from_numpy(x).to(dev), 4, 8, 127, reuse=False)
    y = ops.mx_gemm(xop, wop, wq, None).cpu().numpy()
    ref = O.bfp_linear_int(x, w, None, cfg)
    return np.abs(y - ref).max() / np.abs(ref).max(), y, ref
M, N, K = 256, 256, 256
u = lambda shape: (r.uniform(0.6, 0.99, size=shape) * r.choice([-1, 1], size=shape)).astype(np.float32)
print("all scales equal:", run(u((M, K)), u((N, K)))[0])
# x scales vary per 16-row fragment only
x = u((M, K)); x *= np.repeat(2.0 ** (np.arange(M // 16) % 4), 16)[:, None].astype(np.float32)
print("x scale per row fragment:", run(x, u((N, K)))[0])
w = u((N, K)); w *= np.repeat(2.0 ** (np.arange(N // 16) % 4), 16)[:, None].astype(np.float32)
print("w scale per row fragment:", run(u((M, K)), w)[0])
x = u((M, K)); x *= np.repeat(2.0 ** (np.arange(K // 32) % 4), 32)[None, :].astype(np.float32)
print("x scale per k-group:", run(x, u((N, K)))[0])
e, y, ref = run(u((M, K)), w)
rat = y / ref
print("w-frag case ratio by column fragment:", [float(np.median(rat[:, 16 * j:16 * j + 16])) for j in range(16)])
x = u((M, K)); x *= np.repeat(2.0 ** (np.arange(M // 16) % 4), 16)[:, None].astype(np.float32)
e, y, ref = run(x, u((N, K)))
rat = y / ref
print("x-frag case ratio by row fragment:", [float(np.median(rat[16 * i:16 * i + 16])) for i in range(16)])
print("---- randn cases")
x = r.normal(size=(M, K)).astype(np.float32); w = (r.normal(size=(N, K)) * 0.02).astype(np.float32)
print("randn x, randn w:", run(x, w)[0])
print("randn x, uniform w:", run(x, u((N, K)))[0])
print("uniform x, randn w:", run(u((M, K)), w)[0])
# pairs with different exponents only
x = u((M, K)); x *= np.repeat(2.0 ** (np.arange(K // 16) % 2), 16)[None, :].astype(np.float32)
print("x blocks alternate x1 / x2:", run(x, u((N, K)))[0])
x = u((M, K)); x *= np.repeat(2.0 ** (3 * (np.arange(K // 16) % 2)), 16)[None, :].astype(np.float32)
print("x blocks alternate x1 / x8:", run(x, u((N, K)))[0])
x = (u((M, K)) * 0.3).astype(np.float32)
print("x small mantissas (m <= 2):", run(x * 1.0, u((N, K)))[0])
e, y, ref = run(r.normal(size=(M, K)).astype(np.float32), u((N, K)))
d = np.abs(y - ref); i, j = np.unravel_index(d.argmax(), d.shape); print("worst", i, j, y[i, j], ref[i, j])
