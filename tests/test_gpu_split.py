"""The unquantised layers' fp32-equivalent product on the bf16 MFMA (round 6; include/mi355q.h mi355q_fp32_split_tile, csrc/mi355q_split.hip,
quantized_modules.linear.fp32_linear): the reference keeps the language-model head an fp32 nn.Linear (modeling_llama.py:772,866,
modeling_opt.py:942-944).  The split operand is held bit for bit to a host-side split of the same values; the product to an fp64 product,
at a tolerance the vendor fp32 GEMM itself does not meet everywhere (both errors are measured and compared)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ORDER = {0: "mlhmhh", 1: "mhlhmh"}                      # the six column segments by operand role (csrc/mi355q_split.hip)


def _parts(a):
    h = a.bfloat16().float(); r = a - h
    m = r.bfloat16().float(); l = (r - m).bfloat16().float()
    return {"h": h, "m": m, "l": l}


@pytest.mark.parametrize("rows,K", [(16, 32), (100, 96), (128, 4096), (300, 768), (1, 64)])
@pytest.mark.parametrize("role", [0, 1])
def test_split_operand_is_the_host_split_in_tile_order(rows, K, role):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 7 + K + role)
    x = torch.randn(rows, K, generator=g) * torch.exp(3.0 * torch.randn(rows, K, generator=g))       # a wide range of magnitudes
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-30, -3e30, 1.0 + 2.0 ** -23, 255.0 / 256.0])
    p = _parts(x)
    assert torch.equal((p["h"].double() + p["m"].double() + p["l"].double()).float(), x)             # three parts carry all 24 bits
    want = ops.bf16_tile(torch.cat([p[c] for c in ORDER[role]], dim=1).contiguous().to(dev))
    got = ops.fp32_split_tile(x.to(dev), role)
    assert got.numel() == want.numel()
    # (rows behind `rows` are zeros in the split operand and unwritten in bf16_tile's: compare the pieces' real rows)
    kp6 = 6 * K // 32
    gv = got.view(-1, kp6, 4, 16, 16)[: (rows + 15) // 16].cpu().numpy()
    wv = want.view(-1, kp6, 4, 16, 16)[: (rows + 15) // 16].cpu().numpy()
    full, tail = rows // 16, rows % 16
    assert np.array_equal(gv[:full], wv[:full])
    if tail:
        assert np.array_equal(gv[full, :, :, :tail], wv[full, :, :, :tail])
        assert not gv[full, :, :, tail:].any()
    assert not got.view(-1, kp6, 4, 16, 16)[(rows + 15) // 16:].any()                                # ... up to the 128-row padding


def _errors(M, N, K, seed, bias=False, spread=0.0):
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g) * torch.exp(spread * torch.randn(M, K, generator=g))
    w = torch.randn(N, K, generator=g) * 0.02 * torch.exp(spread * torch.randn(N, K, generator=g))
    b = torch.randn(N, generator=g) if bias else None
    xd, wd = x.to(dev), w.to(dev)
    y = ops.fp32_gemm_split(ops.fp32_split_tile(xd, 0), ops.fp32_split_tile(wd, 1), M, N, K, bias=None if b is None else b.to(dev))
    yv = torch.nn.functional.linear(xd, wd, None if b is None else b.to(dev))
    ref = x.double() @ w.double().t() + (0 if b is None else b.double())
    mag = x.double().abs() @ w.double().abs().t() + (0 if b is None else b.double().abs())          # what the rounding errors scale with
    es = ((y.cpu().double() - ref).abs() / mag)
    ev = ((yv.cpu().double() - ref).abs() / mag)
    assert torch.isfinite(y).all()
    return float(es.max()), float(es.mean()), float(ev.max()), float(ev.mean())


@pytest.mark.parametrize("M,N,K,bias,spread", [(128, 256, 4096, False, 0.0), (256, 1000, 768, True, 0.0), (130, 272, 2048, False, 2.0),
                                               (2048, 512, 4096, True, 0.5), (16, 48, 32, False, 1.0)])
def test_split_product_against_fp64(M, N, K, bias, spread):
    smax, smean, vmax, vmean = _errors(M, N, K, M + N + K, bias, spread)
    # an fp32 product's own bound: every term rounded to 2^-24, accumulated in fp32.  The split keeps the six part products of weight
    # >= 2^-18 exactly and drops <= 3 x 2^-26 of each term; its accumulation is fp32 like the reference GEMM's.
    print(f"split max {smax:.2e} mean {smean:.2e} | vendor fp32 max {vmax:.2e} mean {vmean:.2e}")
    assert smax <= 2e-6 and smean <= 1e-7, (smax, smean)
    assert smean <= 1.5 * vmean + 1e-9 and smax <= 2.0 * vmax + 1e-8, (smax, smean, vmax, vmean)     # no worse than the vendor fp32 GEMM on the same operands


def test_fp32_linear_module_path_cache_and_fallbacks():
    import torch
    import torch.nn as nn
    from mi355q import ops
    from mi355q.quantize import fp32_linear
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    lin = nn.Linear(256, 520, bias=True).to(dev)
    x = torch.randn(2, 96, 256, device=dev)
    ops.vendor_gemm_calls(reset=True)
    with torch.no_grad():
        y = fp32_linear(x, lin)
        ref = (x.double() @ lin.weight.double().t() + lin.bias.double())
        assert y.shape == (2, 96, 520) and float((y.double() - ref).abs().max()) <= 2e-6
        assert ops.vendor_gemm_calls() == {}
        first = lin.__dict__["_mi355q_split_weight"][1]
        fp32_linear(x, lin)
        assert lin.__dict__["_mi355q_split_weight"][1] is first                     # built once ...
        lin.weight.mul_(2.0)                                                         # ... rebuilt when the parameter is written
        y2 = fp32_linear(x, lin)
        assert lin.__dict__["_mi355q_split_weight"][1] is not first
        assert float((y2.double() - (2 * (ref - lin.bias.double()) + lin.bias.double())).abs().max()) <= 4e-6
        # declines: few tokens, "vendor", CPU tensors -- torch's GEMM, counted
        small = fp32_linear(x[:, :8], lin)
        assert torch.allclose(small, torch.nn.functional.linear(x[:, :8], lin.weight, lin.bias))
        fp32_linear(x, lin, "vendor")
        assert sum(ops.vendor_gemm_calls().values()) == 2
    xg = x.clone().requires_grad_(True)                                              # gradients wanted: autograd's F.linear
    fp32_linear(xg, lin).sum().backward()
    assert xg.grad is not None and sum(ops.vendor_gemm_calls(reset=True).values()) == 3


@pytest.mark.parametrize("family", ["llama", "opt"])
def test_model_head_on_the_split_product(family):
    import torch
    from mi355q import harness, ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    q = dict(name="block_fp", bypass=False, is_ptq=True, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
             data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
             bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    if family == "llama":
        cfg = harness.TinyLlamaConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, max_positions=256)
        qc = {"default": q, "rotary_positional_encoding": dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)}
        model = harness.TinyLlamaForCausalLM(cfg, harness.expand_llama_quant_config(qc, 2)).to(dev).eval()
    else:
        cfg = harness.TinyOPTConfig(vocab_size=1000, hidden_size=256, ffn_dim=512, num_layers=2, num_heads=4, max_positions=256)
        model = harness.TinyOPTForCausalLM(cfg, harness.expand_quant_config({"default": q}, 2)).to(dev).eval()
    ids = torch.randint(0, 1000, (1, 256), generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        model(ids)
        ops.vendor_gemm_calls(reset=True)
        logits, loss = model(ids, labels=ids)
        assert ops.vendor_gemm_calls(reset=True) == {}                               # no vendor GEMM left in a forward
        model.mi355q_lm_head = "vendor"
        logits_v, loss_v = model(ids, labels=ids)
        assert sum(ops.vendor_gemm_calls(reset=True).values()) == 1
    assert float((logits - logits_v).abs().max()) <= 2e-5 * float(logits_v.abs().max())
    assert abs(float(loss) - float(loss_v)) <= 1e-5
