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
