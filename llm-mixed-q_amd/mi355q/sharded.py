"""Row-sharded quantised Linear over the GPUs of one node (BASELINE north_star:
"partition the per-layer GEMMs row-wise across the 8 GPUs with RCCL all-gather only for OPT/Llama >= 1.3B").

W [O, K] is split by ROWS (out_features): weight blocks are [1,16] along K, so a row split never cuts a
block and every shard's exponents / mantissas are bit-identical to the unsharded layer's; bias blocks
[16] along O stay whole when O / P is a multiple of the bias block.  The activation is replicated; each
rank quantises it (O(M K), redundant) and computes y[:, shard] with the int8-MFMA path; ONE collective --
all-gather of the fp32 output shards (torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo"
in the CPU tests) -- rebuilds y.  Message size: M x O/P fp32 per rank (4096 x 512 x 4 B = 8 MiB at P = 8).
gather = "quantised" (round 4): where the output only feeds another mi355q Linear through an elementwise step, the rank runs
that step and the CONSUMER's activation quantiser on its own slice and the collective moves the tiled bf16 operand instead
(2 bytes per value; ShardedTiledBf16 below, C ABI mi355q_bf16_gemm_tiled_seg on the consumer's side).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


def shard_bounds(out_features: int, rank: int, world: int, multiple_of: int = 16) -> tuple[int, int]:
    """[lo, hi) of rank's rows.  Shards must be equal and whole bias blocks (all BASELINE shapes are)."""
    if out_features % world:
        raise ValueError(f"out_features={out_features} is not divisible by world size {world}")
    per = out_features // world
    if per % multiple_of:
        raise ValueError(f"shard of {per} rows would cut a bias block of {multiple_of}")
    return rank * per, (rank + 1) * per


class ShardedRows:
    """The gathered output of a RowShardedLinear as the collective leaves it: `buf` [P, M, O/P], rank-major -- row m of the
    [M, O] result is the concatenation of buf[0, m], buf[1, m], ...  A following mi355q Linear reads it in place (its x
    quantiser takes P row segments: ops.block_fp_quantize_aligned_rows(segments=True)); anything else calls dense(), the
    permute copy (M x O fp32 read and written: 20 us at 4096 x 4096) this object exists to avoid."""

    def __init__(self, buf: torch.Tensor, lead: tuple):
        assert buf.ndim == 3 and buf.is_contiguous()
        self.buf, self.lead = buf, tuple(lead)

    @property
    def shape(self):
        return torch.Size((*self.lead, self.buf.shape[0] * self.buf.shape[2]))

    @property
    def device(self):
        return self.buf.device

    def dense(self) -> torch.Tensor:
        p, m, seg = self.buf.shape
        return self.buf.permute(1, 0, 2).reshape(*self.lead, p * seg)


class ShardedTiledBf16:
    """The gathered output of a RowShardedLinear in gather = "quantised" mode: `buf` [P, bytes], rank-major -- segment r is rank
    r's output slice [M, O/P], already passed through the CONSUMER's elementwise step (`pre_applied`: None or "relu") and its
    block_fp activation quantiser into the tiled bf16 operand of the tile GEMM (ops.block_fp_quantize_bf16_tiled: [1,16] blocks
    never straddle a slice, so every rank's blocks are the unsharded layer's).  2 bytes per value cross the links instead of
    4, and each rank quantises 1 / P of the tensor instead of all of it.  Only the consumer it was quantised for can read it
    (`quantiser` = its (width, exponent_width, exponent_bias)): Linear.forward / forward_after on the bf16 flavour of the tile
    GEMM with x in column segments (C ABI mi355q_bf16_gemm_tiled_seg)."""

    def __init__(self, buf: torch.Tensor, lead: tuple, features: int, quantiser: tuple, pre_applied):
        assert buf.ndim == 2 and buf.is_contiguous()
        self.buf, self.lead, self.features, self.quantiser, self.pre_applied = buf, tuple(lead), int(features), tuple(quantiser), pre_applied

    @property
    def shape(self):
        return torch.Size((*self.lead, self.features))

    @property
    def device(self):
        return self.buf.device

    def dense(self):
        raise RuntimeError("mi355q.sharded: a quantised gather holds the consumer's operand, not the fp32 tensor -- shard the "
                           "producer with gather='dense' where something else reads its output")


class RowShardedLinear(nn.Module):
    """Wraps this rank's shard of a (quantised) Linear.  `local` is any module mapping [..., K] -> [..., O/P]."""

    def __init__(self, local: nn.Module, out_features: int, group=None, always_gather: bool = False, gather: str = "dense"):
        super().__init__()
        self.local = local
        self.out_features = out_features
        self.group = group
        self.always_gather = always_gather      # run the collective at world size 1 too (tests / bench on a 1-GPU box)
        assert gather in ("dense", "segments", "quantised")
        self.gather = gather                    # "segments": return the ShardedRows, no permute copy
        # "quantised": the consumer's quantiser parameters and elementwise step (set by shard_model / the caller):
        # (width, exponent_width, exponent_bias) of its data_in, and None or "relu"
        self.consumer_quantiser = None
        self.consumer_pre = None
        # head-sharded attention (round 6, shard_model(heads=True)): a q / k / v projection whose row shard is a whole number of
        # heads hands on this rank's [.., O/P] as it is -- the attention core runs on the rank's own heads and ONE all-gather of
        # its output (gather_heads) feeds out_proj -- instead of one all-gather per projection
        self.keep_local = False

    @classmethod
    def from_full(cls, cls_quantized, linear_fp32: nn.Linear, config: dict, group=None, always_gather: bool = False,
                  gather: str = "dense"):
        """Build this rank's shard from the full-precision layer (every rank holds the checkpoint)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        bias_block = 16
        if linear_fp32.bias is not None and not config.get("bypass", False) and "bias_block_size" in config:
            bs = config["bias_block_size"]
            bias_block = int(bs[-1] if isinstance(bs, (list, tuple)) else bs)
        lo, hi = shard_bounds(linear_fp32.out_features, rank, world, bias_block)
        part = nn.Linear(linear_fp32.in_features, hi - lo, bias=linear_fp32.bias is not None)
        with torch.no_grad():
            part.weight.copy_(linear_fp32.weight[lo:hi])
            if part.bias is not None:
                part.bias.copy_(linear_fp32.bias[lo:hi])
        local = cls_quantized.from_float(part, config).to(linear_fp32.weight.device)
        return cls(local, linear_fp32.out_features, group, always_gather, gather)

    def gather_output(self, y_loc: torch.Tensor, other: torch.Tensor = None):
        """this rank's [.., O/P] -> the layer's [.., O] (or its ShardedRows): the ONE collective of the layer.  `other`:
        (quantised mode, consumer_pre = "silu_mul") the partner projection's local output, same columns"""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if self.gather == "quantised":
            return self._gather_quantised(y_loc, world, other)
        if world == 1 and not (self.always_gather and dist.is_initialized()):
            return ShardedRows(y_loc.detach().reshape(1, -1, y_loc.shape[-1]).contiguous(), y_loc.shape[:-1]) \
                if self.gather == "segments" else y_loc
        lead = y_loc.shape[:-1]
        y2 = y_loc.detach().reshape(-1, y_loc.shape[-1]).contiguous()     # inference path: no autograd through the collective
        gathered, self._gather_buf = self.__dict__.get("_gather_buf"), None
        rank = dist.get_rank(self.group)
        if (gathered is None or gathered.shape != (world * y2.shape[0], y2.shape[1]) or gathered.dtype != y2.dtype
                or y2.data_ptr() != gathered[rank * y2.shape[0]:].data_ptr()):
            # (the local product did not land in the buffer -- a route that allocates its own output: one local copy inside the collective)
            gathered = torch.empty(world * y2.shape[0], y2.shape[1], dtype=y2.dtype, device=y2.device)
        # (in place when y2 IS this rank's segment of `gathered`: sendbuff == recvbuff + rank * count, no local copy)
        dist.all_gather_into_tensor(gathered, y2, group=self.group)          # rank-major: [P * M, O/P]
        COLLECTIVES["all_gather"] += 1
        COLLECTIVES["bytes"] += gathered.numel() * gathered.element_size()
        if self.gather == "segments":
            return ShardedRows(gathered.view(world, y2.shape[0], y2.shape[1]), lead)
        return gathered.view(world, y2.shape[0], y2.shape[1]).permute(1, 0, 2).reshape(*lead, self.out_features)

    def _gather_quantised(self, y_loc: torch.Tensor, world: int, other: torch.Tensor = None):
        """this rank's [.., O/P] -> the consumer's elementwise step (relu; silu(.) * other) -> its block_fp quantiser -> tiled
        bf16 -> ONE all-gather of 2 bytes per value (the collective of the layer)"""
        from . import ops
        if self.consumer_quantiser is None:
            raise RuntimeError("RowShardedLinear(gather='quantised'): consumer_quantiser is not set (shard_model wires it)")
        if y_loc.shape[-1] % 32:
            raise ValueError("gather='quantised': the shard's out_features must be a multiple of 32 (whole K-steps of the consumer)")
        lead = y_loc.shape[:-1]
        y2 = y_loc.detach().reshape(-1, y_loc.shape[-1]).contiguous()
        w, ew, eb = self.consumer_quantiser
        if (self.consumer_pre == "silu_mul") != (other is not None):
            raise RuntimeError("gather='quantised': consumer_pre 'silu_mul' needs the partner projection's local output (grouped_linear)")
        o2 = None if other is None else other.detach().reshape(-1, other.shape[-1]).contiguous()
        pre = None if self.consumer_pre is None else (self.consumer_pre, o2)
        if world == 1 and not (self.always_gather and dist.is_initialized()):
            buf = ops.block_fp_quantize_bf16_tiled(y2, w, ew, eb, reuse=True, pre=pre).reshape(1, -1)
        else:
            # (the operand is written straight into this rank's segment of the gather buffer; the collective runs in place)
            nbytes = ops.bfp_tiled_bytes(y2.shape[0], 2 * y2.shape[1])
            buf = torch.empty(world, nbytes, dtype=torch.int8, device=y2.device)
            mine = ops.block_fp_quantize_bf16_tiled(y2, w, ew, eb, pre=pre, out=buf[dist.get_rank(self.group)])
            dist.all_gather_into_tensor(buf.view(-1), mine, group=self.group)
            COLLECTIVES["all_gather"] += 1
            COLLECTIVES["bytes"] += buf.numel() * buf.element_size()
        return ShardedTiledBf16(buf, lead, self.out_features, self.consumer_quantiser, self.consumer_pre)

    def _aim_at_gather_buffer(self, x):
        """(dense / segments gather) allocate the all-gather buffer FIRST and ask the local layer to store its product in this rank's
        segment of it (quantized_modules.linear._take_out): the collective then runs in place -- round 5: forced world-1 step + 27 ->
        + 5-8 us over the unsharded one, and 1 / P of the bytes less to move at P > 1"""
        self._gather_buf = None
        if self.gather == "quantised" or not dist.is_initialized() or not isinstance(x, (torch.Tensor, ShardedRows)):
            return
        world = dist.get_world_size(self.group)
        dev = x.buf.device if isinstance(x, ShardedRows) else x.device
        if (world == 1 and not self.always_gather) or dev.type != "cuda":
            return
        rows = x.buf.shape[1] if isinstance(x, ShardedRows) else x.numel() // x.shape[-1]
        n_loc = self.local.out_features
        self._gather_buf = torch.empty(world * rows, n_loc, dtype=torch.float32, device=dev)
        rank = dist.get_rank(self.group)
        self.local._out_hint = self._gather_buf[rank * rows:(rank + 1) * rows]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.keep_local:
            return self.local(x)
        self._aim_at_gather_buffer(x)
        try:
            y = self.local(x)
        finally:
            self.local._out_hint = None                       # (a route that did not take it must not find it next time)
        return self.gather_output(y)

    def forward_after(self, x, op, other=None, residual=None):
        """the wrapped layer's fused elementwise step (quantized_modules.linear.forward_after) on this rank's shard; `residual`: added
        behind the gather (the shard's product sees a column slice of it at best: two steps here)"""
        self._aim_at_gather_buffer(x)
        try:
            y = self.local.forward_after(x, op, other)
        finally:
            self.local._out_hint = None
        y = self.gather_output(y)
        if residual is None:
            return y
        return residual + (y.dense() if hasattr(y, "dense") else y)

    def forward_residual(self, x, residual):
        """residual + self(x) (quantized_modules.linear.forward_residual): behind the gather"""
        y = self(x)
        return residual + (y.dense() if hasattr(y, "dense") else y)

    # what the harness reads of a projection
    @property
    def config(self):
        return self.local.config

    @property
    def in_features(self):
        return self.local.in_features


def gather_heads(o_loc: torch.Tensor, group=None, always_gather: bool = False, segments: bool = False):
    """The ONE collective of a head-sharded attention block ("attention bmms could shard by head, independent units,
    no collective until out_proj"; modeling_opt.py:206-328): this rank's heads' output [.., H/P] -> the block's [.., H] in
    front of out_proj / o_proj.  Rank r's columns are heads r nh/P .. (r + 1) nh/P - 1 -- the rows of its q / k / v shards --
    so the rank-major gather buffer [P, M, H/P] IS the column-segmented [M, H].  `segments`: hand on the ShardedRows (an int8
    row-route consumer reads it in place), else the dense tensor."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (always_gather and dist.is_initialized()):
        return o_loc
    lead = o_loc.shape[:-1]
    o2 = o_loc.detach().reshape(-1, o_loc.shape[-1]).contiguous()
    gathered = torch.empty(world * o2.shape[0], o2.shape[1], dtype=o2.dtype, device=o2.device)
    dist.all_gather_into_tensor(gathered, o2, group=group)
    COLLECTIVES["all_gather"] += 1
    COLLECTIVES["bytes"] += gathered.numel() * gathered.element_size()
    if segments:
        return ShardedRows(gathered.view(world, o2.shape[0], o2.shape[1]), lead)
    return gathered.view(world, o2.shape[0], o2.shape[1]).permute(1, 0, 2).reshape(*lead, world * o2.shape[1])


# collectives issued by the sharded layers of this process (tests and the config-4 script read and reset it)
COLLECTIVES = {"all_gather": 0, "bytes": 0}

# the projections of the two harness families that BASELINE's north_star partitions ("the per-layer GEMMs row-wise across the
# 8 GPUs ... for OPT / Llama >= 1.3B"): every quantised Linear of a decoder layer; embeddings, norms, the attention core
# (heads replicated: SURVEY 8e) and the unquantised lm_head stay whole on every rank
SHARDED_PROJECTIONS = {
    "opt": (("self_attn", ("q_proj", "k_proj", "v_proj", "out_proj")), (None, ("fc1", "fc2"))),
    "llama": (("self_attn", ("q_proj", "k_proj", "v_proj", "o_proj")), (None, ("gate_proj", "up_proj", "down_proj"))),
}


def _quantised_gather_fits(consumer, features: int, group) -> bool:
    """the consumer (fc2 / down_proj) can read a quantised gather: block_fp with [1,16] blocks along in_features, activation AND
    weight widths <= 9 (both operands go to the bf16 tile GEMM: wider values would be rounded there -- ADVICE r4), whole 64-byte
    K-steps per rank's slice, and the consumer itself configured for the fused elementwise step (the harness decides from the
    CONSUMER's config whether it calls forward_after: ADVICE r4)"""
    c = consumer.local.config if isinstance(consumer, RowShardedLinear) else consumer.config
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    bs = c.get("data_in_block_size")
    return (c.get("name") == "block_fp" and not c.get("bypass", False) and c.get("is_ptq", False) and 2 <= c["data_in_width"] <= 9
            and 2 <= c["weight_width"] <= 9 and c.get("mi355q_fused_activation", False)
            and list(bs)[-1] == 16 and all(b == 1 for b in list(bs)[:-1]) and features % (32 * world) == 0)


def shard_model(model: nn.Module, group=None, always_gather: bool = False, gather: str = "dense", heads: bool = True) -> nn.Module:
    """Row-shard every quantised Linear of a harness model (mi355q.harness.TinyOPTForCausalLM / TinyLlamaForCausalLM) in place:
    rank r keeps rows [r O/P, (r + 1) O/P) of each projection's weight and bias -- never cutting a [1,16] weight block or a
    [16] bias block -- quantises and packs only those, and one all-gather per projection rebuilds its output.  Call it on the
    full-precision model (before its first forward quantises the weights in place), on every rank, with identical weights.
    The reference runs models of this size by LAYER placement instead (cli/eval_perplexity.py:66-75: accelerate's
    infer_auto_device_map over decoder layers); this is the row partition BASELINE.json's north_star asks for.
    `gather`: "dense" -- every projection hands on [.., O]; "segments" -- fc1 (OPT) hands its rank-major ShardedRows to fc2's
    quantiser as it lies (no permute copy), everything else dense (the attention core and the gated product need [.., O]);
    "quantised" -- fc1 (OPT) applies fc2's relu and fc2's activation quantiser to ITS OWN slice and gathers the tiled bf16
    operand (2 bytes per value instead of 4, 1 / P of the quantiser's work per rank); fc2 then multiplies on the bf16 flavour
    of the tile GEMM with x in column segments (results as the per-block route's: exact products, fp32 accumulation).  Llama:
    gate / up (grouped) -> silu(gate) * up and down_proj's quantiser on the rank's slice, ONE gather for the pair.
    `heads` (round 6, default): where a rank's row shard of q / k / v is a whole number of heads (num_heads % P == 0) the
    attention core runs on THOSE heads -- the three projections are not gathered at all -- and one all-gather of the attention
    output stands in front of out_proj / o_proj (gather_heads): OPT 6 -> 4 collectives a layer, Llama 7 -> 5, a third of the
    gathered bytes and (P - 1) / P of the attention work gone; the [1,16] blocks of all four attention operands lie inside one
    head, so every head's result is the unsharded model's, bit for bit."""
    from .quantize.quantized_modules.linear import _LinearBase
    family = "llama" if hasattr(model.layers[0], "gate_proj") else "opt"
    world_ = dist.get_world_size(group) if dist.is_initialized() else 1
    for layer in model.layers:
        for owner_name, names in SHARDED_PROJECTIONS[family]:
            owner = layer if owner_name is None else getattr(layer, owner_name)
            for name in names:
                lin = getattr(owner, name)
                if isinstance(lin, RowShardedLinear):
                    continue
                if not isinstance(lin, _LinearBase):
                    raise TypeError(f"{name}: expected a quantised Linear of the registry, found {type(lin).__name__}")
                if not lin.weight_requires_quantisation and not lin.bypass:
                    if lin.config.get("mi355q_weight_storage", "int8") == "packed" and getattr(lin, "_master", None) is None:
                        raise RuntimeError(f"{name}: its weights were quantised and packed when they arrived on the GPU "
                                           "(mi355q_weight_storage = 'packed' packs on arrival): shard the model on the CPU, before "
                                           ".to(device), or keep master weights (mi355q_keep_master)")
                    raise RuntimeError(f"{name}: shard the model before its first forward quantises the weights in place")
                shim = nn.Linear(lin.in_features, lin.out_features, bias=lin.bias is not None, device="meta")
                shim.weight, shim.bias = lin.weight, lin.bias               # (the full-precision parameters, no copy)
                def _cfg_of(mod):
                    return mod.local.config if isinstance(mod, RowShardedLinear) else mod.config
                seg = (gather == "segments" and name == "fc1" and lin.config.get("mi355q_fused_activation", False)
                       and _cfg_of(getattr(owner, "fc2")).get("mi355q_fused_activation", False))
                qnt = (gather == "quantised" and name == "fc1" and lin.config.get("mi355q_fused_activation", False)
                       and _quantised_gather_fits(getattr(owner, "fc2"), lin.out_features, group))
                # Llama: gate / up -> silu(gate) * up -> down_proj (needs the grouped launch: the pair is gathered as one)
                # (decided for the PAIR: both quantised or both dense -- the gated product needs its partner's shard on the rank)
                gated = (gather == "quantised" and name in ("gate_proj", "up_proj")
                         and all(_cfg_of(getattr(owner, n)).get("mi355q_fused_activation", False)
                                 and _cfg_of(getattr(owner, n)).get("mi355q_grouped_linear", False) for n in ("gate_proj", "up_proj"))
                         and _quantised_gather_fits(getattr(owner, "down_proj"), lin.out_features, group))
                qnt = qnt or gated
                wrapped = RowShardedLinear.from_full(type(lin), shim, lin.config, group, always_gather,
                                                     "segments" if seg else ("quantised" if qnt else "dense"))
                if qnt:
                    nxt = owner.down_proj if gated else owner.fc2
                    c2 = nxt.local.config if isinstance(nxt, RowShardedLinear) else nxt.config
                    wrapped.consumer_quantiser = (c2["data_in_width"], c2["data_in_exponent_width"], c2["data_in_exponent_bias"])
                    wrapped.consumer_pre = "silu_mul" if gated else "relu"
                if (heads and owner_name == "self_attn" and name in ("q_proj", "k_proj", "v_proj")
                        and owner.nh % world_ == 0 and (lin.out_features // world_) % owner.hd == 0):
                    wrapped.keep_local = True
                    owner.mi355q_head_shard = (group, always_gather)
                setattr(owner, name, wrapped)
    model.mi355q_sharded = True
    return model
