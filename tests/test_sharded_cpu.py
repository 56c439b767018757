"""N > 1 path on CPU: world_size-2 gloo processes run the row-sharded Linear (collective plumbing), and
the oracle shows that a row shard's quantised weights / bias equal the slices of the unsharded ones."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_q):
    import torch
    import torch.distributed as dist
    import mi355q.quantize as Q
    from mi355q.sharded import RowShardedLinear
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                     # every rank holds the same full-precision layer
        full = torch.nn.Linear(64, 96)
        cfg = {"name": "block_fp", "bypass": True, "data_in_width": 6, "weight_width": 6}
        cls = Q.get_quantized_cls("linear", cfg)
        sh = RowShardedLinear.from_full(cls, full, cfg)
        assert sh.local.out_features == 96 // world
        x = torch.randn(3, 5, 64)
        y = sh(x)
        ref = full(x)
        # the same gathered in the collective's own rank-major layout (no permute copy): dense() re-assembles it, a following
        # bypassed Linear takes it through dense() as well
        seg = RowShardedLinear.from_full(cls, full, cfg, gather="segments")(x)
        nxt = torch.nn.Linear(96, 16)
        nq = cls.from_float(nxt, cfg)
        ok = (tuple(seg.buf.shape) == (world, 15, 96 // world) and seg.shape == y.shape and torch.equal(seg.dense(), y)
              and torch.allclose(nq(seg), nxt(ref), rtol=1e-5, atol=1e-6))
        out_q.put((rank, bool(torch.allclose(y, ref, rtol=1e-5, atol=1e-6)) and bool(ok), tuple(y.shape)))
    except Exception as e:          # report instead of leaving the parent waiting on the queue
        out_q.put((rank, False, repr(e)))
    finally:
        dist.destroy_process_group()


def test_row_sharded_linear_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, shape in res:
        assert same, f"rank {rank}: gathered output differs from the unsharded layer"
        assert shape == (3, 5, 96)


def test_shard_bounds_errors():
    from mi355q.sharded import shard_bounds
    assert shard_bounds(4096, 3, 8) == (1536, 2048)
    with pytest.raises(ValueError):
        shard_bounds(100, 0, 8)
    with pytest.raises(ValueError):
        shard_bounds(96, 0, 4, 16)          # 24 rows per rank would cut a 16-wide bias block


@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_shards_quantise_identically(world):
    """the claim the sharding rests on: quantising a row shard == slicing the quantised full tensor"""
    from oracle import np_oracle as O
    r = np.random.default_rng(world)
    w = (r.normal(size=(128 * world, 96)) * 0.02).astype(np.float32)
    b = (r.normal(size=(128 * world,)) * 0.02).astype(np.float32)
    full = O.bfp_encode(w, 6, 8, 127, [1, 16], False)
    bq = O.block_fp_quantize(b, 6, 8, 127, [16], False)
    nb = 96 // 16
    for k in range(world):
        lo, hi = k * 128, (k + 1) * 128
        part = O.bfp_encode(w[lo:hi], 6, 8, 127, [1, 16], False)
        assert np.array_equal(part.exp, full.exp[lo * nb:hi * nb])
        assert np.array_equal(part.mant, full.mant[lo * nb:hi * nb])
        assert np.array_equal(O.block_fp_quantize(b[lo:hi], 6, 8, 127, [16], False), bq[lo:hi])


def _full_cfg(**extra):
    d = dict(name="block_fp", bypass=True, is_ptq=True,
             data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
             weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
             bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    d.update(extra)
    return d


def _build_model(family, torch, harness):
    torch.manual_seed(0)                          # every rank holds the same full-precision model
    if family == "opt":
        c = harness.TinyOPTConfig(vocab_size=96, hidden_size=64, ffn_dim=128, num_layers=2, num_heads=4, max_positions=32)
        return harness.TinyOPTForCausalLM(c, harness.expand_quant_config(_full_cfg(), 2))
    c = harness.TinyLlamaConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_layers=2, num_heads=4, max_positions=32)
    return harness.TinyLlamaForCausalLM(c, harness.expand_llama_quant_config(_full_cfg(), 2))


def _model_worker(rank, world, port, out_q):
    import torch
    import torch.distributed as dist
    from mi355q import harness, sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        # (the Llama family's rotary function quantises whatever `bypass` says, like the reference's, and there is no CPU
        #  quantiser: the Llama harness is sharded in the GPU test, tests/test_gpu_sharded.py)
        for family in ("opt",):
            ids = torch.randint(0, 96, (2, 16), generator=torch.Generator().manual_seed(3))
            ref, ref_loss = _build_model(family, torch, harness)(ids, labels=ids)
            # heads = True (default, round 6): q / k / v keep this rank's two heads, ONE all-gather of the attention output in front
            # of out_proj -- 4 collectives a layer; heads = False: heads replicated, one all-gather per projection -- 6
            for heads, per_layer in ((True, 4), (False, 6 if family == "opt" else 7)):
                model = sharded.shard_model(_build_model(family, torch, harness), heads=heads)
                lin = model.layers[0].self_attn.q_proj
                assert isinstance(lin, sharded.RowShardedLinear) and lin.local.out_features == 64 // world
                assert lin.keep_local == heads and (getattr(model.layers[0].self_attn, "mi355q_head_shard", None) is not None) == heads
                sharded.COLLECTIVES.update(all_gather=0, bytes=0)
                got, loss = model(ids, labels=ids)
                res.append((f"{family} heads={heads}", bool(torch.allclose(got, ref, rtol=1e-5, atol=1e-6)), abs(float(loss) - float(ref_loss)) < 1e-5,
                            sharded.COLLECTIVES["all_gather"] == 2 * per_layer))
        out_q.put((rank, res))
    except Exception as e:
        out_q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_row_sharded_model_gloo_world2():
    """row (h): a sharded MODEL -- every quantised Linear of the OPT harness split over two gloo ranks -- gives the unsharded
    logits and loss: with the attention core head-sharded (each rank two of the four heads, one all-gather of the attention
    output in front of out_proj: four collectives a layer) and with the heads replicated (one all-gather per projection: six)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, r in res:
        assert isinstance(r, list), f"rank {rank}: {r}"
        for family, logits_ok, loss_ok, coll_ok in r:
            assert logits_ok and loss_ok, f"rank {rank} {family}: sharded model differs from the unsharded one"
            assert coll_ok, f"rank {rank} {family}: unexpected number of all-gathers"


def test_shard_model_refuses_a_model_that_already_quantised_its_weights():
    import torch
    from mi355q import harness, sharded
    m = _build_model("opt", torch, harness)
    m.layers[0].fc1.bypass = False
    m.layers[0].fc1.weight_requires_quantisation = False
    with pytest.raises(RuntimeError):
        sharded.shard_model(m)


def _bfp_cfg(xw=6, ww=6, **extra):
    c = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=xw, data_in_exponent_width=8, data_in_exponent_bias=127,
             data_in_block_size=[1, 16], weight_width=ww, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
             bias_width=ww, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    c.update(extra)
    return c


def test_quantised_gather_needs_a_consumer_that_can_take_it():
    """a quantised gather feeds the bf16 tile GEMM, so the CONSUMER's weights must be exact in bf16 too (width <= 9:
    a search result with weight_width 12 would be rounded silently), and the consumer itself must be configured for the fused
    elementwise step -- the harness decides from ITS config whether it calls forward_after"""
    import torch
    import mi355q.quantize as Q
    from mi355q.sharded import _quantised_gather_fits
    mk = lambda cfg: Q.get_quantized_cls("linear", cfg)(256, 64, bias=True, config=cfg)
    assert _quantised_gather_fits(mk(_bfp_cfg(6, 6, mi355q_fused_activation=True)), 256, None)
    assert not _quantised_gather_fits(mk(_bfp_cfg(6, 12, mi355q_fused_activation=True)), 256, None)      # wide weights
    assert not _quantised_gather_fits(mk(_bfp_cfg(12, 6, mi355q_fused_activation=True)), 256, None)      # wide activations
    assert not _quantised_gather_fits(mk(_bfp_cfg(6, 6)), 256, None)                                     # consumer without the knob
    assert not _quantised_gather_fits(mk(_bfp_cfg(6, 6, mi355q_fused_activation=True, bypass=True)), 256, None)


def test_shard_model_decides_gathers_from_both_ends_of_the_pair():
    """per-layer overrides in which producer and consumer disagree about the fused elementwise step (a search TOML
    entry without mi355q_* keys), or only one of gate / up qualifies: every such projection gathers densely instead of handing
    the harness an object it then applies F.relu / F.silu to"""
    import torch
    from mi355q import harness as H
    from mi355q.sharded import RowShardedLinear, shard_model
    knobs = dict(mi355q_fused_activation=True, mi355q_grouped_linear=True)
    base = _bfp_cfg(6, 6, bypass=True, **knobs)
    # OPT: fc2 of layer 0 without the knob -> fc1 of layer 0 dense, layer 1 as asked
    qc = H.expand_quant_config(dict(base), 2)
    qc["model_layer_0"]["fc2"] = {k: v for k, v in qc["model_layer_0"]["fc2"].items() if not k.startswith("mi355q_")}
    cfg = H.TinyOPTConfig(vocab_size=64, hidden_size=64, ffn_dim=128, num_layers=2, num_heads=2, max_positions=32)
    for gather in ("segments", "quantised"):
        m = shard_model(H.TinyOPTForCausalLM(cfg, qc), gather=gather)
        assert m.layers[0].fc1.gather == "dense", gather
        # (bypassed layers cannot take a quantised gather either way; segments need nothing of the arithmetic)
        assert m.layers[1].fc1.gather == ("segments" if gather == "segments" else "dense")
    # Llama: up_proj of layer 0 without the grouped knob -> gate AND up dense
    lq = H.expand_llama_quant_config(_bfp_cfg(6, 6, **knobs), 2)
    lq["model_layer_0"]["mlp"]["up_proj"] = {k: v for k, v in lq["model_layer_0"]["mlp"]["up_proj"].items() if k != "mi355q_grouped_linear"}
    lcfg = H.TinyLlamaConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_layers=2, num_heads=2, max_positions=32)
    m = shard_model(H.TinyLlamaForCausalLM(lcfg, lq), gather="quantised")
    assert m.layers[0].gate_proj.gather == "dense" and m.layers[0].up_proj.gather == "dense"
    assert m.layers[1].gate_proj.gather == "quantised" and m.layers[1].up_proj.gather == "quantised"
    assert isinstance(m.layers[0].down_proj, RowShardedLinear)
