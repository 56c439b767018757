"""The row-sharded quantised Linear on the GPU with backend "nccl" (= RCCL): world size 1 always, world size 2 when
the box has two GPUs.  Quantised path (W4A4 and W6A6), result == the unsharded layer bit for bit (a row split never
cuts a [1,16] weight block)."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg(width):
    return dict(name="block_fp", is_ptq=True, bypass=False,
                data_in_width=width, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
                weight_width=width, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
                bias_width=width, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import mi355q.quantize as Q
        from mi355q import ops
        from mi355q.sharded import RowShardedLinear
        from oracle import np_oracle as O
        res = []
        for width, (K, N, M) in ((4, (2048, 8192, 256)), (6, (1024, 1024, 300))):     # OPT-1.3B fc1 shape, W4A4 (config 4)
            cfg = _cfg(width)
            torch.manual_seed(7)
            full = torch.nn.Linear(K, N)
            x = (torch.randn(2, M // 2, K) * torch.exp(torch.randn(2, M // 2, 1))).to(dev)
            cls = Q.get_quantized_cls("linear", cfg)
            # (always_gather: all_gather_into_tensor over RCCL runs at world size 1 too -- the collective, its rank-major
            #  layout and the un-permute are exercised on a 1-GPU box, counted below)
            sh = RowShardedLinear.from_full(cls, full.to(dev), cfg, always_gather=True)
            calls, real = [], dist.all_gather_into_tensor
            dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
            try:
                y = sh(x)
            finally:
                dist.all_gather_into_tensor = real
            assert len(calls) == 1, "the RCCL all-gather did not run"
            whole = cls.from_float(full, cfg).to(dev)
            y_ref = whole(x)
            same = bool(torch.equal(y, y_ref))
            ref = O.bfp_linear_int(x[0, :16].cpu().numpy(), full.weight.detach().cpu().numpy(),
                                   full.bias.detach().cpu().numpy(), cfg)
            err = float((y[0, :16].cpu() - torch.from_numpy(ref)).abs().max() / abs(ref).max())
            # gathered in the collective's own layout and handed to the next Linear as it lies (ShardedRows: the x quantiser reads
            # P row segments): the same bits as from the re-assembled tensor
            torch.manual_seed(8)
            nxt = cls.from_float(torch.nn.Linear(N, 512), cfg).to(dev)
            z_ref = nxt(y_ref)
            z_ref = nxt(y_ref)                       # (second call: packed weights, row-aligned route decided)
            seg = RowShardedLinear.from_full(cls, full.to(dev), cfg, always_gather=True, gather="segments")(x)
            same = same and tuple(seg.buf.shape) == (world, M, N // world) and bool(torch.equal(seg.dense(), y_ref))
            took = []
            real_q = ops.block_fp_quantize_aligned_rows
            ops.block_fp_quantize_aligned_rows = lambda *a, **k: (took.append(k.get("segments", False)), real_q(*a, **k))[1]
            try:
                z = nxt(seg)
            finally:
                ops.block_fp_quantize_aligned_rows = real_q
            same = same and bool(torch.equal(z, z_ref)) and (took == [True] or nxt._align_mode != "rows")
            res.append((width, same, err, tuple(y.shape), sh.local._packed is not None))
        q.put((rank, res))
    except Exception as e:
        import traceback
        q.put((rank, "".join(traceback.format_exception(e))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_row_sharded_linear_nccl(world):
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, res in out:
        assert not isinstance(res, str), res
        for width, same, err, shape, packed in res:
            assert same, f"rank {rank} W{width}: sharded output differs from the unsharded layer"
            assert err < 1e-5, (rank, width, err)
            assert packed, "the shard did not take the int8 path"


def _model_cfg(width, mixed, knobs):
    d = _cfg(width)
    if knobs:
        d.update(mi355q_fused_attention=True, mi355q_grouped_linear=True, mi355q_fused_activation=True, mi355q_token_major_output=True)
    cfg = {"default": d}
    if mixed:       # per-layer widths, as a search result writes them (experiments/emnlp/configs/search/opt_1.3b_sst2.toml)
        cfg["model_layer_1"] = {"self_attn": {"q_proj": dict(d, data_in_width=6, weight_width=5, bias_width=5)},
                                "fc2": dict(d, data_in_width=5, weight_width=3, bias_width=3)}
    return cfg


def _model_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from mi355q import harness, sharded
        res = []
        # heads: the attention core on the rank's own heads, one all-gather of its output in front of out_proj / o_proj (round 6:
        # OPT 4 collectives a layer, Llama 5) -- or heads replicated and q, k, v gathered one by one (6 / 7)
        # (last field: the rotary embedding applied by the attention pack launch -- on the rank's own heads when head-sharded)
        for family, width, mixed, knobs, gather, heads, rotary in (("opt", 4, True, False, "dense", True, False), ("opt", 4, True, True, "segments", True, False),
                                                                   ("llama", 6, False, True, "dense", True, False), ("opt", 4, True, True, "dense", False, False),
                                                                   ("llama", 6, False, False, "dense", True, False), ("llama", 6, False, True, "dense", False, False),
                                                                   ("llama", 6, False, True, "dense", True, True), ("llama", 6, False, True, "dense", False, True)):
            def build():
                torch.manual_seed(11)
                if family == "opt":
                    c = harness.TinyOPTConfig(vocab_size=512, hidden_size=512, ffn_dim=2048, num_layers=2, num_heads=8, max_positions=256)
                    m = harness.TinyOPTForCausalLM(c, harness.expand_quant_config(_model_cfg(width, mixed, knobs), 2))
                else:
                    lc = _model_cfg(width, mixed, knobs)
                    lc["default"]["mi355q_fused_rotary"] = rotary
                    lc["rotary_positional_encoding"] = dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)
                    c = harness.TinyLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1536, num_layers=2, num_heads=8, max_positions=256)
                    m = harness.TinyLlamaForCausalLM(c, harness.expand_llama_quant_config(lc, 2))
                with torch.no_grad():
                    for p in m.parameters():
                        if p.ndim == 2:
                            p.mul_(torch.exp(0.5 * torch.randn(p.shape[0], 1)))
                return m.to(dev).eval()
            ids = torch.randint(0, 512, (1, 256), generator=torch.Generator().manual_seed(5)).to(dev)
            with torch.no_grad():
                whole = build()
                for _ in range(2):
                    ref, ref_loss = whole(ids, labels=ids)
                model = sharded.shard_model(build(), always_gather=True, gather=gather, heads=heads)
                for _ in range(2):                      # (first forward packs, the second runs the settled routes)
                    sharded.COLLECTIVES.update(all_gather=0, bytes=0)
                    got, loss = model(ids, labels=ids)
            n_proj = (4 if family == "opt" else 5) if heads else (6 if family == "opt" else 7)
            packed = model.layers[0].self_attn.q_proj.local._packed is not None
            assert model.layers[0].self_attn.q_proj.keep_local == heads
            res.append((family, knobs, f"{gather} heads={heads} rotary={rotary}", bool(torch.equal(got, ref)), float(loss) == float(ref_loss),
                        sharded.COLLECTIVES["all_gather"] == 2 * n_proj, packed))
        q.put((rank, res))
    except Exception as e:
        import traceback
        q.put((rank, "".join(traceback.format_exception(e))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_row_sharded_model_nccl(world):
    """row (h): every quantised Linear of a harness model row-sharded (sharded.shard_model), real RCCL all-gathers (at world size
    1 too), QUANTISED layers -- W4A4 with mixed per-layer widths (BASELINE config 4's kind) and W6A6 Llama: logits and loss equal
    the unsharded model's bit for bit; with the attention core head-sharded (round 6: each rank its own heads, ONE all-gather of
    the attention output in front of out_proj / o_proj -- OPT 6 -> 4 collectives a layer, Llama 7 -> 5) and with the heads
    replicated (one all-gather per projection)"""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, res in out:
        assert not isinstance(res, str), res
        for family, knobs, gather, same, loss_same, coll, packed in res:
            assert same and loss_same, f"rank {rank} {family} knobs={knobs} gather={gather}: sharded model differs from the unsharded one"
            assert coll, f"rank {rank} {family}: unexpected number of all-gathers"
            assert packed, "the shards did not take the int8 path"


def _quantised_gather_worker(rank, world, port, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import mi355q.quantize as Q
        from mi355q import harness, sharded
        from mi355q.sharded import RowShardedLinear, ShardedTiledBf16
        from oracle import np_oracle as O
        res = {}
        # layer level: fc1 (sharded, gather="quantised": relu + fc2's quantiser on the rank's own slice, 2 bytes per value
        # gathered) -> fc2 on the gathered operand, against the oracle's exact contraction of relu(fc1(x)) and against the
        # unsharded pair
        cfg1, cfg2 = _cfg(4), dict(_cfg(4), data_in_width=5, weight_width=6)
        torch.manual_seed(3)
        K, F_, N, M = 512, 2048, 512, 200
        l1, l2 = torch.nn.Linear(K, F_), torch.nn.Linear(F_, N)
        x = (torch.randn(2, M // 2, K) * torch.exp(torch.randn(2, M // 2, 1))).to(dev)
        cls = Q.get_quantized_cls("linear", cfg1)
        fc1 = RowShardedLinear.from_full(cls, l1.to(dev), cfg1, always_gather=True, gather="quantised")
        fc1.consumer_quantiser, fc1.consumer_pre = (cfg2["data_in_width"], cfg2["data_in_exponent_width"], cfg2["data_in_exponent_bias"]), "relu"
        fc2 = cls.from_float(l2, cfg2).to(dev)
        sharded.COLLECTIVES.update(all_gather=0, bytes=0)
        with torch.no_grad():
            g = fc1(x)
            y = fc2.forward_after(g, "relu")
            y = fc2.forward_after(fc1(x), "relu")
        res["gathered"] = isinstance(g, ShardedTiledBf16) and tuple(g.buf.shape)[0] == world and sharded.COLLECTIVES["all_gather"] == 2
        res["bytes_per_value"] = sharded.COLLECTIVES["bytes"] / 2 / (M * F_)          # 2 (+ row padding of the tiles)
        w1, w2 = cls.from_float(l1, cfg1).to(dev), cls.from_float(l2, cfg2).to(dev)
        with torch.no_grad():
            h = w1(x)
            y_ref = w2(torch.relu(h))
        href = O.bfp_linear_int(x.reshape(-1, K).cpu().numpy(), l1.weight.detach().cpu().numpy(), l1.bias.detach().cpu().numpy(), cfg1)
        res["fc1_exact"] = bool(np.abs(h.reshape(-1, F_).cpu().numpy() - href).max() <= 1e-5 * np.abs(href).max())
        yo = O.bfp_linear_int(np.maximum(h.reshape(-1, F_).cpu().numpy(), 0), l2.weight.detach().cpu().numpy(), l2.bias.detach().cpu().numpy(), cfg2)
        res["err_vs_oracle"] = float(np.abs(y.reshape(-1, N).cpu().numpy() - yo).max() / np.abs(yo).max())
        res["err_vs_unsharded"] = float((y - y_ref).abs().max() / y_ref.abs().max())
        try:
            g.dense()
            res["dense_raises"] = False
        except RuntimeError:
            res["dense_raises"] = True
        # model level: OPT harness, W4A4 mixed, every knob on; fc1 -> fc2 through the quantised gather
        def build():
            torch.manual_seed(11)
            c = harness.TinyOPTConfig(vocab_size=512, hidden_size=512, ffn_dim=2048, num_layers=2, num_heads=8, max_positions=256)
            m = harness.TinyOPTForCausalLM(c, harness.expand_quant_config(_model_cfg(4, True, True), 2))
            return m.to(dev).eval()
        ids = torch.randint(0, 512, (1, 256), generator=torch.Generator().manual_seed(5)).to(dev)
        with torch.no_grad():
            whole = build()
            for _ in range(2):
                ref, ref_loss = whole(ids, labels=ids)
            model = sharded.shard_model(build(), always_gather=True, gather="quantised")
            for _ in range(2):
                sharded.COLLECTIVES.update(all_gather=0, bytes=0)
                got, loss = model(ids, labels=ids)
        res["model_modes"] = [model.layers[i].fc1.gather for i in range(2)]
        res["model_collectives"] = sharded.COLLECTIVES["all_gather"]
        res["model_logits_err"] = float((got - ref).abs().max() / ref.abs().max())
        res["model_loss_err"] = abs(float(loss) - float(ref_loss))
        # Llama: gate / up (one grouped launch) -> silu(gate) * up -> down_proj's quantiser, all on the rank's own slice
        def build_llama():
            torch.manual_seed(12)
            lc = _model_cfg(6, False, True)
            lc["default"]["mi355q_fused_norm"] = False
            lc["rotary_positional_encoding"] = dict(name="integer", bypass=False, data_in_width=8, data_in_frac_width=7)
            c = harness.TinyLlamaConfig(vocab_size=512, hidden_size=512, intermediate_size=1536, num_layers=2, num_heads=8, max_positions=256)
            return harness.TinyLlamaForCausalLM(c, harness.expand_llama_quant_config(lc, 2)).to(dev).eval()
        with torch.no_grad():
            whole = build_llama()
            for _ in range(2):
                ref, ref_loss = whole(ids, labels=ids)
            model = sharded.shard_model(build_llama(), always_gather=True, gather="quantised")
            for _ in range(2):
                sharded.COLLECTIVES.update(all_gather=0, bytes=0)
                got, loss = model(ids, labels=ids)
        res["llama_modes"] = [model.layers[i].gate_proj.gather for i in range(2)] + [model.layers[i].down_proj.gather for i in range(2)]
        res["llama_collectives"] = sharded.COLLECTIVES["all_gather"]          # attention heads, o, (gate + up as one), down per layer
        res["llama_logits_err"] = float((got - ref).abs().max() / ref.abs().max())
        res["llama_loss_err"] = abs(float(loss) - float(ref_loss))
        q.put((rank, res))
    except Exception as e:
        import traceback
        q.put((rank, "".join(traceback.format_exception(e))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_quantised_gather_nccl(world):
    """gather = "quantised": a rank applies the consumer's relu and activation quantiser to its OWN output slice and the ranks
    all-gather the tiled bf16 operand (2 bytes per value instead of 4; 1 / P of the quantiser's work each); the consumer's product
    runs on the bf16 tile GEMM with x in column segments.  Layer pair against the oracle's exact contraction; a sharded OPT
    harness model against the unsharded one (another route for fc2: equal up to fp32 summation order, then W4 re-rounding)"""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_quantised_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, res in out:
        assert not isinstance(res, str), res
        assert res["gathered"] and res["dense_raises"] and res["fc1_exact"], res
        assert 2.0 <= res["bytes_per_value"] <= 2.7, res
        assert res["err_vs_oracle"] <= 4e-6 and res["err_vs_unsharded"] <= 4e-6, res
        assert res["model_modes"] == ["quantised", "quantised"] and res["model_collectives"] == 8, res     # (heads + out_proj + fc1 + fc2 a layer)
        assert res["model_logits_err"] <= 5e-2 and res["model_loss_err"] <= 5e-3, res
        assert res["llama_modes"] == ["quantised", "quantised", "dense", "dense"] and res["llama_collectives"] == 8, res   # (heads, o, gate + up as one, down)
        assert res["llama_logits_err"] <= 5e-2 and res["llama_loss_err"] <= 5e-3, res
