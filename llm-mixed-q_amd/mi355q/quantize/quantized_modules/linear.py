"""Quantised `nn.Linear` classes with the reference's contract
(`models/quantize/quantized_modules/linear.py:31-101, 113-203`): subclass of nn.Linear, ctor
`(in_features, out_features, bias, device, dtype, config)`, attributes `config, bypass, is_ptq,
weight_requires_quantisation, x_quantizer, w_quantizer, b_quantizer`, `from_float`, `__repr__`.

MI355X path of `LinearBlockFP` in PTQ mode (the hot path):
  first forward : W <- Qw(W), b <- Qb(b) in place (as the reference, linear.py:66-70) AND the int8
                  mantissas + uint8 shared exponents of W are packed once and kept on the module;
  every forward : x -> quantise+pack kernel -> exponent-align -> int8-MFMA block GEMM -> fp32 y (+ b), no fake-quant
                  tensor and no fp32 GEMM.  Operands no row window fits (post-activation inputs, weights with outlier
                  input channels) keep every block's exponent: x -> quantise straight into tiled bf16 -> the bf16
                  flavour of the same tile GEMM (values exact in bf16, products exact in fp32).
Formats whose contraction is not an int8 dot (block_minifloat, block_log, exotic block shapes,
QAT) quantise with the HIP fake-quant kernels and contract with the stock fp32 GEMM on the GPU.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ..quantizers import QUANTIZER_MAP

# config-key suffixes each quantiser takes, per operand prefix (linear.py:113-203)
_PARAMS = {
    "block_fp": dict(width="width", exponent_width="exponent_width", exponent_bias="exponent_bias",
                     block_size="block_size"),
    "block_minifloat": dict(width="width", exponent_width="exponent_width",
                            exponent_bias_width="exponent_bias_width", block_size="block_size"),
    "block_log": dict(width="width", exponent_bias_width="exponent_bias_width", block_size="block_size"),
    "integer": dict(width="width", frac_width="frac_width"),
    "minifloat_ieee": dict(width="width", exponent_width="exponent_width", exponent_bias="exponent_bias"),
    "minifloat_denorm": dict(width="width", exponent_width="exponent_width", exponent_bias="exponent_bias"),
    # the reference passes exponent_width to log_quantizer, which rejects it
    "log": dict(width="width", exponent_width="exponent_width", exponent_bias="exponent_bias"),
}
_BLOCKED = ("block_fp", "block_minifloat", "block_log")


def _make_quantizer(arith: str, config: dict, prefix: str, skip_first_dim: bool):
    kw = {arg: config[f"{prefix}_{suffix}"] for arg, suffix in _PARAMS[arith].items()}
    if arith in _BLOCKED:
        kw["skip_first_dim"] = skip_first_dim
    elif arith == "integer":
        kw["is_signed"] = True
    return partial(QUANTIZER_MAP[arith], **kw)


def _capturing_graph() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _bf16_planes(t):
    """fp32 [rows, K] -> (hi, lo) with hi + lo = t to 2^-17 relative: hi = t rounded to bf16, lo the remainder (rounded to bf16
    when it is tiled).  A gradient is not a quantised value: one bf16 plane would cost it 2^-9."""
    hi = t.to(torch.bfloat16).to(torch.float32)
    return hi, t - hi


class _TileLinear(torch.autograd.Function):
    """y = x_q . W_q^T + b_q on the bf16 tile GEMM (operands exact in bf16, fp32 accumulation), and its backward as two more
    products on the same kernel: dX = dY . W_q, dW = dY^T . x_q, db = sum dY (reference: torch.autograd through F.linear,
    quantized_modules/linear.py:72-76; the quantisers' straight-through estimators stay where they are).  dY enters as two
    bf16 planes (hi + lo): four launches per backward."""

    @staticmethod
    def forward(ctx, xq, wq, bq):
        K, N = wq.shape[1], wq.shape[0]
        x2 = xq.reshape(-1, K).contiguous()
        y = ops.bf16_gemm_tiled(ops.bf16_tile(x2), ops.bf16_tile(wq.contiguous()), x2.shape[0], N, K, bq)
        ctx.save_for_backward(x2, wq)
        ctx.lead, ctx.has_bias = xq.shape[:-1], bq is not None
        return y.reshape(*xq.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, wq = ctx.saved_tensors
        M, K, N = x2.shape[0], x2.shape[1], wq.shape[0]
        dy2 = dy.reshape(M, N).contiguous().to(torch.float32)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:                              # dX [M, K] = dY [M, N] . (W_q^T [K, N])^T
            wt = ops.bf16_tile(wq.t().contiguous())
            hi, lo = _bf16_planes(dy2)
            dx = ops.bf16_gemm_tiled(ops.bf16_tile(hi), wt, M, K, N)
            dx += ops.bf16_gemm_tiled(ops.bf16_tile(lo), wt, M, K, N)
            dx = dx.reshape(*ctx.lead, K)
        if ctx.needs_input_grad[1]:                              # dW [N, K] = dY^T [N, M] . (x_q^T [K, M])^T
            xt = ops.bf16_tile(x2.t().contiguous())
            hi, lo = _bf16_planes(dy2.t().contiguous())
            dw = ops.bf16_gemm_tiled(ops.bf16_tile(hi), xt, N, K, M)
            dw += ops.bf16_gemm_tiled(ops.bf16_tile(lo), xt, N, K, M)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        return dx, dw, db


class _LinearBase(nn.Linear):
    arith: str = None

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=None,
                 config: dict = None) -> None:
        super().__init__(in_features, out_features, bias, device, dtype)
        self.config = config
        self.bypass = config.get("bypass", False)
        self.is_ptq = config.get("is_ptq", False)
        self.weight_requires_quantisation = True if self.is_ptq else False
        self.x_quantizer = self.w_quantizer = self.b_quantizer = None
        self._packed = None          # (aligned W operand, its tiled mantissas, weight._version, bias._version)
        # exponent alignment of the packed operands (an implementation knob, not part of the reference config):
        # "rows" = one exponent per row (row-scale int8 GEMM), "blocks" = every block keeps its exponent (bf16 tile GEMM),
        # "auto" = rows when the weights and the first activations fit it, re-checked on a doubling schedule
        # (the alignment per 256-value group, "groups", went in round 5 with its int32-chain kernel)
        self.align = config.get("mi355q_align", "auto")
        self._align_mode, self._calls, self._row_overflows = None, 0, 0
        self._w_bf16 = None          # (tiled bf16 weights, weight._version) of the per-block-exponent route
        self._mx_w, self._mx_calls, self._mx_version = None, 0, -1      # W4A4 on the MX scaled MFMA: the weights' operand
        # implementation knobs next to "mi355q_align" (not part of the reference config):
        #   mi355q_weight_storage = "packed": keep the weights at rest as width-bit mantissas + one byte per block
        #       (width + 0.5 bits per value, ops.PackedWeights) and stream them into a shared scratch operand every forward;
        #   mi355q_keep_master = True: keep the fp32 weights / bias the layer was given, so that requantize() can quantise
        #       them again (other widths included) without a checkpoint reload (the search loop)
        #   mi355q_mixed = "auto" (default) / False: layers whose activations (or weights) carry OUTLIER CHANNELS -- block columns that
        #       lie outside their rows' exponent window in a large share of the rows -- split in_features into two classes of block
        #       columns and run ONE launch of the mixed contraction (ops.bfp_gemm_mixed: class 0 on the int8 MFMA, class 1 on the
        #       bf16 MFMA) instead of the whole layer on the bf16 flavour (round 6)
        self._mixed = None           # dict(classes, wa0, w1): the column split and the weights' two operands
        self._w_packed = None
        self._pending_flavour = None
        self._master = None
        self._fp32_released = False
        self._x_cap = {"rows_post": ops.ROW_BUCKET_CAP_MAX, "blocks": ops.ROW_NO_ALIGN}.get(self.align, ops.ACTIVATION_BUCKET_CAP)
        if not self.bypass:
            self._setup_quantizers(config)
        if self._packs_at_load():
            self.register_load_state_dict_post_hook(lambda module, incompatible: module._loaded_new_weights())

    # ---- mi355q_weight_storage = "packed": the weights are quantised and packed WHEN THEY ARRIVE -- when the module reaches
    #      the GPU (from_float(...).to(device), model.to(device)) and when a state dict is loaded into it -- not at the first
    #      forward: a model loaded for inference holds width + 0.5 bits per weight before it has seen a token (SURVEY 8f.2,
    #      quantized_modules/linear.py:66-70 is where the reference does the same work, at the first forward)
    def _packs_at_load(self) -> bool:
        return (not self.bypass and self.is_ptq and self.arith == "block_fp"
                and self.config.get("mi355q_weight_storage", "int8") == "packed")

    def _pack_if_arrived(self):
        if self._packs_at_load() and self.weight_requires_quantisation and self.weight.is_cuda and self.weight.numel():
            self.pack_now()

    def _loaded_new_weights(self):
        with torch.no_grad():
            self.weight_requires_quantisation = True          # (fresh fp32 values: quantise them again)
            self._packed, self._w_bf16, self._w_packed, self._pending_flavour, self._mixed = None, None, None, None, None
            if self._master is not None:
                self._master = (self.weight.detach().clone(), None if self.bias is None else self.bias.detach().clone())
            self._pack_if_arrived()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self.__dict__.get("config") is not None:
            pend = self.__dict__.get("_pending_flavour")
            if pend is not None and pend[0].device != self.weight.device:
                self._pending_flavour = None        # (moved again before its first forward: the repack route decides, forward())
            with torch.no_grad():
                self._pack_if_arrived()
        return out

    def _setup_quantizers(self, config: dict):
        self.x_quantizer = _make_quantizer(self.arith, config, "data_in", True)
        self.w_quantizer = _make_quantizer(self.arith, config, "weight", False)
        self.b_quantizer = _make_quantizer(self.arith, config, "bias", False) if self.bias is not None else None

    # -- int8 block GEMM eligibility ------------------------------------------------------
    def _int8_plan(self, x: torch.Tensor):
        """(x_mbits, w_mbits, x_bias, w_bias) when the contraction is an int8 x int8 block dot on
        the MFMA path: block_fp both sides, [1,16] blocks along in_features, widths <= 8."""
        if self.arith != "block_fp" or not x.is_cuda or x.dtype != torch.float32 or x.ndim < 2:
            return None
        key = (x.ndim, x.shape[-2] if x.ndim == 3 else 0, id(self.config))
        cached = self.__dict__.get("_plan_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        plan = self._int8_plan_uncached(x)
        self.__dict__["_plan_cache"] = (key, plan)
        return plan

    def _int8_plan_uncached(self, x: torch.Tensor):
        c, K = self.config, self.in_features
        if K % 64 or not (2 <= c["data_in_width"] <= 8 and 2 <= c["weight_width"] <= 8):
            return None
        if not (1 <= c["data_in_exponent_width"] <= 8 and 1 <= c["weight_exponent_width"] <= 8):
            return None
        xs = [1, K] if x.ndim == 2 else [1, x.shape[-2], K]
        if x.ndim > 3:
            return None
        if ops.resolve_blocking(xs, c["data_in_block_size"], True)[3:] != (1, 16):
            return None
        if ops.resolve_blocking([self.out_features, K], c["weight_block_size"], False)[3:] != (1, 16):
            return None
        xb, wb = c["data_in_exponent_bias"], c["weight_exponent_bias"]
        xb = 2 ** (c["data_in_exponent_width"] - 1) - 1 if xb in (None, "none", "None") else xb
        wb = 2 ** (c["weight_exponent_width"] - 1) - 1 if wb in (None, "none", "None") else wb
        if xb < 0 or wb < 0:
            return None                  # packed operands store biased uint8 exponent codes: non-negative biases only
        return c["data_in_width"] - 1, c["weight_width"] - 1, xb, wb

    @torch.no_grad()
    def _quantise_weights_once(self, pack: bool, x_sample=None):
        """linear.py:66-70 plus the one-off packing of the int8 operand"""
        c = self.config
        if c.get("mi355q_keep_master", False) and self._master is None:
            self._master = (self.weight.detach().clone(), None if self.bias is None else self.bias.detach().clone())
        self._mx_w = None
        if pack and self._mx_config_ok():
            # W4A4 on the MX scaled MFMA (csrc/mi355q_mx.hip): the operand from the RAW weights (block_fp is not idempotent), kept
            # only if every 32-group of W fits the format (one host read, here where the weights are packed anyway)
            mw = ops.block_fp_quantize_mx(self.weight.data.contiguous(), c["weight_width"], c["weight_exponent_width"], c["weight_exponent_bias"],
                                          reuse=False, keep_source=False)
            if _capturing_graph() or int(mw.bad[0]) == 0:
                self._mx_w = mw
        if pack:
            wq, wm, we = ops.block_fp_quantize(self.weight.data, c["weight_width"], c["weight_exponent_width"],
                                               c["weight_exponent_bias"], c["weight_block_size"], False,
                                               want_packed=True)
            self.weight.copy_(wq)
        else:
            self.weight.copy_(self.w_quantizer(self.weight.data))
        if self.bias is not None:
            self.bias.copy_(self.b_quantizer(self.bias.data))
        self.weight_requires_quantisation = False
        self._mx_version = self.weight._version
        if pack:
            self._pack_operands(wm, we, x_sample)
            # packed on arrival (no activation seen yet, `_pack_if_arrived`): the integers stay until the first forward has
            # decided which flavour the layer's activations call for (`_finalise_packed_flavour`)
            self._pending_flavour = (wm, we) if x_sample is None and self._packs_at_load() else None

    def _pack_operands(self, wm, we, x_sample):
        c = self.config
        self._align_weights(wm, we, self._choose_align_mode(wm, we, x_sample))
        versions = (self.weight._version, None if self.bias is None else self.bias._version)
        if self._mixed is not None:
            self._mixed["version"] = versions
        # "packed": every layer at rest as width-bit mantissas + a byte per block, expanded per forward; "hybrid" (round 6): only the
        # layers on the per-block-exponent route -- whose resident operand is 16 bits a value, and whose expand pass is a third of a
        # Llama layer's -- while the row-scale layers keep their int8 operands (8.5 bits a value) and every fused path that needs them
        storage = c.get("mi355q_weight_storage", "int8")
        packed_storage = storage == "packed" and self._align_mode == "rows"
        self._w_packed = None
        if self._uses_bf16_route():
            # per-block exponents: the quantised weights (already in .weight) as tiled bf16; the int8 operand is not
            # needed on this route (the exact-integer blockwise kernel, mi355q_blocks_gemm = "int8", keeps it)
            if packed_storage or (storage == "hybrid" and self._align_mode == "rows"):
                self._w_packed = ops.pack_block_exponent_weights(wm, we, c["weight_width"], self._weight_bias_value())
                self._packed = (None, self._w_packed.packed, *versions)
            else:
                self._w_bf16 = (ops.bf16_tile(self.weight.data), self.weight._version)
                self._packed = (None, self._w_bf16[0], *versions)
        elif packed_storage and self._x_cap != ops.ROW_NO_ALIGN:
            self._w_packed = ops.pack_row_aligned_weights(wm, we, self._packed[0], c["weight_width"], self._weight_bias_value())
            self._packed = (None, self._w_packed.packed, *versions)

    @torch.no_grad()
    def _finalise_packed_flavour(self, x):
        """first forward of a layer that packed its weights on arrival: the activation-dependent half of the route decision
        (`_choose_align_mode`), from the kept integers; then they go"""
        wm, we = self._pending_flavour
        self._pending_flavour = None
        if not self._fp32_released and self._packed is not None:
            self._x_cap = {"rows_post": ops.ROW_BUCKET_CAP_MAX, "blocks": ops.ROW_NO_ALIGN}.get(self.align, ops.ACTIVATION_BUCKET_CAP)
            self._pack_operands(wm, we, x)

    def _weight_bias_value(self):
        c = self.config
        wb = c["weight_exponent_bias"]
        return 2 ** (c["weight_exponent_width"] - 1) - 1 if wb in (None, "none", "None") else wb

    def _choose_align_mode(self, wm, we, x_sample):
        if self.align == "groups":
            raise ValueError('mi355q_align = "groups" was removed in round 5 (use "auto", "rows", "rows_post" or "blocks")')
        if not ops.row_align_supported(self.in_features):
            # contractions past the row format's 16384 (Llama-30B/65B down_proj): every block keeps its exponent
            self._x_cap = ops.ROW_NO_ALIGN
            return "rows"
        if self.align in ("rows", "rows_post", "blocks"):
            return "rows"
        c = self.config
        # one-off host reads at pack time: overflow words and the fullest exception bucket of either operand.  Rows pay
        # off while a 256 x 256 tile's entries (x bucket + w bucket) fit the GEMM's in-LDS add-back.
        wa = ops.bfp_align_rows(wm, we, c["weight_width"] - 1, self._weight_bias_value())
        w_over, w_max = ops.row_list_fill(wa.sparse, self.out_features)
        if w_over != 0 or w_max > ops.ROW_TILE_ENTRIES_FAST:
            # weights whose exception blocks do not fit a tile's LDS add-back (outlier input channels put one in every
            # row): the outlier block columns as class 1 of the mixed contraction if that leaves a class 0 that fits, else no
            # alignment -- the blockwise / bf16 product does not care how exponents are distributed
            if not self._try_mixed(wm, we, x_sample):
                self._x_cap = ops.ROW_NO_ALIGN
            return "rows"
        if x_sample is not None:
            # activations: the GEMM's in-LDS add-back while a tile's entries fit it; otherwise (post-activation inputs:
            # hundreds of exception blocks per 256 rows after a ReLU, no usable row window at all after a SiLU gate) no
            # alignment -- measured faster than the row post-pass wherever that one applies (tools/timing/time_linear_modes.py;
            # "rows_post" remains available explicitly)
            xa = ops.block_fp_quantize_aligned_rows(x_sample.reshape(-1, self.in_features), c["data_in_width"],
                                                    c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                                    bucket_cap=ops.ROW_BUCKET_CAP_MAX)
            x_over, x_max = ops.row_list_fill(xa.sparse, xa.rows, xa.list_cap)
            # (a 128-row tile carries about half of its 256-row bucket's activation entries; 15 % margin for the busier
            # half.  Measured at 2048 x 4096 -> 4096, tools/timing/time_exception_density.py: the row-scale route wins up to a
            # fullest activation bucket of ~60 there, the per-block route beyond ~85)
            # Between 48 and 96 entries a tile forms its vectors behind the K loop: still ahead of the per-block route where
            # that one runs 256-row tiles (2048 x 4096 -> 11008: 163 vs 208 us at a fullest bucket of 87), behind it on
            # 128-row tiles (2048 x 4096 -> 4096: 100 vs 91 us).
            tile_rows = ops.gemm_tile_rows(xa.rows, self.out_features)
            n_tile = w_max + int(x_max * (0.5 * 1.15 if tile_rows == 128 else 1.0) + 0.999)
            fits = x_over == 0 and (n_tile <= ops.ROW_TILE_ENTRIES_FAST or
                                    (tile_rows == 256 and n_tile <= ops.ROW_TILE_ENTRIES_SLOW - 8))
            self._x_cap = ops.ROW_BUCKET_CAP if fits else ops.ROW_NO_ALIGN
            if not fits and self._try_mixed(wm, we, x_sample):
                self._x_cap = ops.ROW_BUCKET_CAP
        return "rows"

    @staticmethod
    def _outlier_share(codes, spare: int):
        """codes [rows, blocks] (biased block exponents): per block column, the share of the rows in which the block lies
        outside a window of spare + 1 exponents centred on the row's median exponent"""
        c = codes.to(torch.int16)
        med = c.median(dim=1, keepdim=True).values
        out = (c < med - spare // 2) | (c > med + (spare - spare // 2))
        return out.float().mean(0)

    def _try_mixed(self, wm, we, x_sample) -> bool:
        """The mixed contraction for a layer whose rows fit no exponent window AS A WHOLE (README.md:9-11 of the reference: a few
        input channels tens of times larger than the rest): block columns that are outliers in more than a quarter of the rows --
        of the sample activations or of the weights -- become class 1 (tiled bf16, every block its own exponent); if they are at
        most half of the columns and the remaining class 0 fits the row-scale route's exception buckets on both operands, the layer
        runs ONE launch with class 0 on the int8 MFMA (ops.bfp_gemm_mixed) instead of all of it on the bf16 flavour.  Decided once
        per packing from the first activations, like the row / block decision itself; results never depend on it."""
        c, K = self.config, self.in_features
        self._mixed = None
        if (c.get("mi355q_mixed", "auto") in (False, "off", None) or self.align != "auto" or x_sample is None or K % 128 or K < 512
                or not ops.row_align_supported(K) or c.get("mi355q_weight_storage", "int8") == "packed"
                or c["data_in_width"] > 8 or c["weight_width"] > 8 or not x_sample.is_cuda or x_sample.dtype != torch.float32):
            return False
        nb = K // 16
        x2 = x_sample.reshape(-1, K)
        _, _, xe = ops.block_fp_quantize(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"], [1, 16], True,
                                         want_fake=False, want_packed=True)
        share = torch.maximum(self._outlier_share(xe.view(-1, nb), 8 - c["data_in_width"]),
                              self._outlier_share(we.view(-1, nb), 8 - c["weight_width"]))
        n1 = int((share > 0.25).sum())
        if n1 == 0:
            return False
        n1 = -(-n1 // 8) * 8                                  # whole pairs of 64-byte K-steps in both classes
        if n1 > nb // 2 or nb - n1 < 16:
            return False
        blocks1 = torch.topk(share, n1).indices.cpu().tolist()
        classes = ops.ColumnClasses(K, blocks1, x2.device)
        N = self.out_features
        wa0 = ops.bfp_align_rows(wm.view(N, K)[:, classes.cols0].contiguous(), we.view(N, nb)[:, classes.blocks0].contiguous(),
                                 c["weight_width"] - 1, self._weight_bias_value())
        w_over, w_max = ops.row_list_fill(wa0.sparse, N)
        if w_over != 0 or w_max > ops.ROW_TILE_ENTRIES_FAST:
            return False
        x0, _ = ops.block_fp_quantize_classes(x2, classes, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                              bucket_cap=ops.ROW_BUCKET_CAP_MAX)
        x_over, x_max = ops.row_list_fill(x0.sparse, x0.rows, x0.list_cap)
        if x_over != 0 or w_max + x_max > ops.ROW_TILE_ENTRIES_FAST:
            return False
        # (the weights' class-1 operand: the fake-quantised values already in .weight, as tiled bf16)
        w1 = ops.bf16_tile(self.weight.data[:, classes.cols1].contiguous())
        self._mixed = dict(classes=classes, wa0=wa0, w1=w1, version=None)
        return True

    def _uses_bf16_route(self) -> bool:
        c = self.config
        return (self._align_mode == "rows" and self._x_cap == ops.ROW_NO_ALIGN and self.in_features % 32 == 0
                and (c.get("mi355q_blocks_gemm", "bf16") == "bf16" or not ops.row_align_supported(self.in_features))
                and c["data_in_width"] <= 9 and c["weight_width"] <= 9)

    def _align_weights(self, wm, we, mode):
        c = self.config
        self._align_mode = mode
        if not ops.row_align_supported(self.in_features):     # (bf16 route only: `_pack_operands` sets the operand)
            self._packed = (None, None, self.weight._version, None if self.bias is None else self.bias._version)
            return
        wa = ops.bfp_align_rows(wm, we, c["weight_width"] - 1, self._weight_bias_value())
        self._packed = (wa, wa.tiled, self.weight._version, None if self.bias is None else self.bias._version)

    def _packed_is_current(self) -> bool:
        p = self._packed
        return (p is not None and p[2] == self.weight._version and p[1].device == self.weight.device
                and (self.bias is None or p[3] == self.bias._version))

    @torch.no_grad()
    def _repack_quantised_weights(self, x_sample):
        """the module was moved to another device after its first forward: pack again from .weight, which holds the
        quantised values -- a quantised tensor re-encodes exactly only through a cast (the quantiser is not idempotent,
        SURVEY 8a quirk 7), so the route from here on is the per-block-exponent one (tiled bf16).  Weights edited in
        place after the first forward are NOT repacked: like the reference they are used as they are (fp32 GEMM) until
        requantize()."""
        if self.in_features % 32 or self.config["weight_width"] > 9 or self.config["data_in_width"] > 9:
            return False
        self._align_mode, self._x_cap, self._mixed = "rows", ops.ROW_NO_ALIGN, None
        self._w_bf16 = (ops.bf16_tile(self.weight.data.contiguous()), self.weight._version)
        self._packed = (None, self._w_bf16[0], self.weight._version, None if self.bias is None else self.bias._version)
        return True

    def requantize(self, config: dict = None):
        """Search-loop helper (the reference rebuilds the model and reloads the checkpoint for every trial,
        search/search.py:753-763): make the next forward quantise and pack the weights again.  With
        config["mi355q_keep_master"] the fp32 weights / bias the layer was first given are restored here, so the caller
        reloads nothing; `config` switches the layer to another quantisation config (other widths) on the way.  Without a
        master copy the caller loads new fp32 values into .weight / .bias first (after a first forward they hold
        quantised values)."""
        if config is not None:
            keep = {k: v for k, v in self.config.items() if k.startswith("mi355q_") and k not in config}
            self.config = dict(config, **keep)
            self.bypass = self.config.get("bypass", False)
            self.is_ptq = self.config.get("is_ptq", False)
            self.align = self.config.get("mi355q_align", "auto")
            if not self.bypass:
                self._setup_quantizers(self.config)
        if self._master is not None:
            with torch.no_grad():
                if self._fp32_released:
                    self.weight.data = torch.empty_like(self._master[0])
                    self._fp32_released = False
                self.weight.copy_(self._master[0])
                if self.bias is not None:
                    self.bias.copy_(self._master[1])
        elif self._fp32_released:
            raise RuntimeError("mi355q: requantize() needs fp32 weights: they were released and no master copy is kept")
        self.weight_requires_quantisation = True if self.is_ptq else False
        self._packed, self._align_mode, self._calls, self._row_overflows = None, None, 0, 0
        self._w_bf16, self._w_packed, self._pending_flavour, self._mixed = None, None, None, None
        self._x_cap = {"rows_post": ops.ROW_BUCKET_CAP_MAX, "blocks": ops.ROW_NO_ALIGN}.get(self.align, ops.ACTIVATION_BUCKET_CAP)

    @torch.no_grad()
    def pack_now(self, x_sample=None):
        """Quantise and pack the weights NOW instead of at the first forward (a loader calls this right after it filled
        .weight / .bias: pack at load).  Without an activation sample the activation route is decided at
        the first forward."""
        if self.bypass or not self.is_ptq or not self.weight_requires_quantisation:
            return self
        probe = x_sample if x_sample is not None else torch.empty(1, self.in_features, device=self.weight.device)
        self._quantise_weights_once(pack=self._int8_plan(probe) is not None, x_sample=x_sample)
        return self

    def release_fp32_weight(self):
        """Drop the fp32 copy of the (already packed) weights: what stays resident is the packed operand -- with
        config["mi355q_weight_storage"] = "packed" width + 0.5 bits per value.  .weight becomes an empty Parameter;
        only the packed routes work afterwards (requantize() brings the weights back if a master copy is kept)."""
        if self._packed is None:
            raise RuntimeError("mi355q: nothing packed yet (run a forward or pack_now() first)")
        # the MX route's exact in-launch fallback reads the fp32 weights: that operand goes with them
        self._mx_w = None
        self.weight.data = torch.empty(0, dtype=self.weight.dtype, device=self.weight.device)
        self._packed = (self._packed[0], self._packed[1], self.weight._version, self._packed[3])
        if self._w_bf16 is not None:
            self._w_bf16 = (self._w_bf16[0], self.weight._version)
        self._fp32_released = True
        return self

    def weight_storage_bits(self) -> float:
        """bits per weight value of what the layer keeps packed (excluding the fp32 Parameter, see release_fp32_weight)"""
        if self._w_packed is not None:
            return self._w_packed.bits_per_value()
        if self._packed is None:
            return 32.0
        n = self.out_features * self.in_features
        if self._packed[0] is not None:
            wa = self._packed[0]
            return 8.0 * (wa.tiled.numel() + wa.exp.numel()) / n
        return 8.0 * self._packed[1].numel() / n

    def forward(self, x):
        from ...sharded import ShardedRows
        if isinstance(x, ShardedRows):
            # the gathered output of a row-sharded layer, still in the collective's rank-major layout: the row-aligned int8
            # route reads it in place; every other route takes the re-assembled tensor
            if (not self.bypass and self.is_ptq and not self.weight_requires_quantisation and self._packed_is_current()
                    and self._pending_flavour is None
                    and self._align_mode == "rows" and not self._uses_bf16_route()):
                plan = self._int8_plan(x.buf[0])
                if plan is not None:
                    with torch.no_grad():
                        return self._forward_int8(x, plan)
            x = x.dense()
        from ...sharded import ShardedTiledBf16
        if isinstance(x, ShardedTiledBf16):
            if x.pre_applied is not None:
                raise RuntimeError(f"mi355q: a quantised gather that applied {x.pre_applied!r} reached forward(); call forward_after")
            return self._forward_quantised_gather(x)
        if self.bypass:
            ops.count_vendor_gemm("linear.bypass")
            return F.linear(x, self.weight, self.bias)
        if self.is_ptq:
            plan = self._int8_plan(x)
            differentiated = torch.is_grad_enabled() and x.requires_grad      # (someone wants d/dx through this layer)
            with torch.no_grad():
                if self.weight_requires_quantisation:
                    self._quantise_weights_once(pack=plan is not None, x_sample=x)
                elif self._pending_flavour is not None:
                    self._finalise_packed_flavour(x)
                p = self._packed
                if (plan is not None and p is not None and p[1].device != self.weight.device
                        and p[2] == self.weight._version and (self.bias is None or p[3] == self.bias._version)):
                    self._repack_quantised_weights(x)        # the module was moved: same (quantised) values, new device
                if plan is not None and self._mx_w is not None and self._mx_takes(x):
                    return self._forward_mx(x)
                if plan is not None and self._packed_is_current():
                    return self._forward_int8(x, plan)
                if not differentiated and self._values_exact_in_bf16(x):
                    return self._forward_bf16_values(x)
                if not differentiated and self._padded_block_fp_ok(x):
                    return self._forward_block_fp_padded(x)
                x = self.x_quantizer(x)
            ops.count_vendor_gemm("linear.ptq_fallthrough")
            return F.linear(x, self.weight, self.bias)
        x = self.x_quantizer(x)
        w = self.w_quantizer(self.weight)
        bias = self.b_quantizer(self.bias) if self.bias is not None else None
        if self._qat_on_tile_gemm(x):
            return _TileLinear.apply(x, w, bias)
        ops.count_vendor_gemm("linear.qat_fp32")
        return F.linear(x, w, bias)

    def _qat_on_tile_gemm(self, xq) -> bool:
        """QAT (is_ptq = False, linear.py:72-76: x, W and b re-quantised every call, straight-through gradients): the product
        F.linear(x_q, W_q, b_q) and its two backward products on the bf16 flavour of the tile GEMM instead of the library's fp32
        GEMM at a sixteenth of its rate -- when the quantised values are exact in bf16 (block_fp of <= 9 bits, minifloats of
        <= 7 mantissa bits, powers of two) and every contraction length is a whole number of 64-byte K-steps.
        config["mi355q_qat_gemm"] = "fp32" keeps F.linear."""
        c = self.config
        if c.get("mi355q_qat_gemm", "bf16") not in ("bf16", "bf16_always") or not (xq.is_cuda and xq.dtype == torch.float32 and self.weight.dtype == torch.float32):
            return False
        if self.arith == "block_fp":
            if not (2 <= c["data_in_width"] <= 9 and 2 <= c["weight_width"] <= 9):
                return False
        elif not self._values_exact_in_bf16(xq):
            return False
        M = xq.numel() // self.in_features
        if not (self.in_features % 32 == 0 and self.out_features % 32 == 0 and M % 32 == 0 and M > 0):
            return False
        # three products + the tiling / plane-split launches around them: ahead of the fp32 library GEMM from ~2^34 multiply-adds a
        # product (profiles/r05_qat_gemm.jsonl: 2048 x 1024 x 4096 0.97-1.09x, 2048 x 4096 x 4096 2.0x, 512 x 1024 x 4096 0.4-0.6x);
        # mi355q_qat_gemm = "bf16_always" takes it regardless (tests)
        return c.get("mi355q_qat_gemm", "bf16") == "bf16_always" or M * self.in_features * self.out_features >= (1 << 34)

    # -- W4A4 on the MX scaled matrix instruction ------------------------------------------------------------------------
    def _mx_config_ok(self) -> bool:
        """config["mi355q_mx"]: "auto" (default) -- block_fp operands of <= 4 bits each (every mantissa exact in FP6 e2m3 with
        three exponents of reach inside a 32-group; at 5 bits the reach is two and Gaussian data already trips it), [1,16] blocks
        along in_features, in_features % 128 == 0, launches of >= 192 tiles of 256 x 256 (below that the small-tile int8 kernel
        wins: profiles/r05_mx_w4a4.txt); True -- every launch that fits the format (<= 5 bits); False -- never"""
        c = self.config
        knob = c.get("mi355q_mx", "auto")
        if knob in (False, "off", None) or self.arith != "block_fp" or not self.is_ptq or self.bypass:
            return False
        wmax = 5 if knob is True else 4
        return (ops.mx_supported(self.in_features, c["data_in_width"], c["weight_width"]) and c["data_in_width"] <= wmax
                and c["weight_width"] <= wmax and self.weight.is_cuda and self.weight.dtype == torch.float32
                and c.get("mi355q_weight_storage", "int8") != "packed" and not getattr(self, "_fp32_released", False))

    def _mx_takes(self, x) -> bool:
        if not (x.is_cuda and x.dtype == torch.float32 and 2 <= x.ndim <= 3) or self._mx_w is None or self._fp32_released:
            return False
        if self._mx_w.c16.device != x.device or self._mx_version != self.weight._version:
            return False
        if self.config.get("mi355q_mx", "auto") is True:
            return True
        M = x.numel() // self.in_features
        return -(-M // 256) * -(-self.out_features // 256) >= 192

    def _forward_mx(self, x):
        c = self.config
        x2 = x.reshape(-1, self.in_features).contiguous()
        xop = ops.block_fp_quantize_mx(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"])
        y = ops.mx_gemm(xop, self._mx_w, self.weight.data, self.bias)
        # a 32-group of the activations that does not fit the format sends the launch to its exact (slow) route: look at the
        # flag on a doubling schedule of calls and leave this route if it is ever up (never while a graph is being recorded)
        self._mx_calls += 1
        if self._mx_calls & (self._mx_calls - 1) == 0 and not _capturing_graph():
            if int(xop.bad[0]) != 0:
                self._mx_w = None
        return y.reshape(*x.shape[:-1], self.out_features)

    def _values_exact_in_bf16(self, xq) -> bool:   # (xq: the layer's input, quantised or not: only its placement matters)
        """block_minifloat / block_log PTQ layers (linear.py:145-203; likewise the un-blocked minifloat_ieee / minifloat_denorm
        and integer ones, :104-110, 206-263): the fake-quantised values -- minifloats with at most 7 mantissa bits, signed
        powers of two, fixed point of at most 9 bits -- are exact in bf16 and a product of two of them exact in fp32, so
        `F.linear(x_q, W_q, b_q)` is the bf16 flavour of the tile GEMM (fp32 accumulation, fp32 output) instead of a
        library fp32 GEMM at a seventh of its rate.  config["mi355q_values_gemm"] = "fp32" keeps F.linear."""
        c = self.config
        if c.get("mi355q_values_gemm", "bf16") != "bf16":
            return False
        if self.arith in ("block_minifloat", "minifloat_ieee", "minifloat_denorm"):     # <= 7 mantissa bits
            if not all(0 <= c[f"{p}_width"] - c[f"{p}_exponent_width"] - 1 <= 7 for p in ("data_in", "weight")):
                return False
        elif self.arith == "integer":                          # fixed point of <= 9 bits: <= 8 significant bits
            if not all(2 <= c[f"{p}_width"] <= 9 for p in ("data_in", "weight")):
                return False
        elif self.arith != "block_log":                        # (signed powers of two)
            return False
        return (xq.is_cuda and xq.dtype == torch.float32 and self.weight.dtype == torch.float32 and xq.ndim >= 2
                and self.in_features % 32 == 0 and not self.weight_requires_quantisation)

    @torch.no_grad()
    def _forward_quantised_gather(self, x):
        """x: sharded.ShardedTiledBf16 -- this layer's operand as the ranks quantised and gathered it (one tiled bf16 segment
        per rank): the bf16 flavour of the tile GEMM with x in column segments against the tiled quantised weights"""
        c = self.config
        if (self.arith != "block_fp" or self.bypass or not self.is_ptq
                or x.quantiser != (c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"]) or x.features != self.in_features):
            raise RuntimeError("mi355q: this quantised gather was prepared for another layer / quantiser")
        if not 2 <= c["weight_width"] <= 9:
            raise RuntimeError(f"mi355q: a quantised gather feeds the bf16 tile GEMM, whose operands hold block_fp values of at most 9 "
                               f"bits exactly; this layer's weight_width is {c['weight_width']} (shard it with gather='dense')")
        if self.weight_requires_quantisation:
            self._quantise_weights_once(pack=False)          # (linear.py:66-70: weights and bias quantised in place; no int8 operand needed)
        if self._w_bf16 is None or self._w_bf16[1] != self.weight._version or self._w_bf16[0].device != x.device:
            if getattr(self, "_fp32_released", False):
                raise RuntimeError("mi355q: the fp32 weights were released; this layer cannot take the bf16 route any more")
            self._w_bf16 = (ops.bf16_tile(self.weight.data), self.weight._version)
        M = 1
        for d in x.lead:
            M *= int(d)
        P = x.buf.shape[0]
        y = ops.bf16_gemm_tiled(x.buf if P > 1 else x.buf[0], self._w_bf16[0], M, self.out_features, self.in_features, self.bias,
                                segments=P)
        return y.reshape(*x.lead, self.out_features)

    def _padded_block_fp_ok(self, x) -> bool:
        """block_fp layers whose in_features is a multiple of the block (16) but not of the tile kernels' K-step (64): the
        contraction is padded with all-zero blocks -- they quantise to zeros and add nothing -- and runs on the bf16 flavour of
        the tile GEMM (a block_fp value of width <= 9 is exact in bf16, products exact in fp32) instead of dropping to the
        library fp32 GEMM.  config["mi355q_pad_k"] = False keeps F.linear."""
        c, K = self.config, self.in_features
        if self.arith != "block_fp" or not c.get("mi355q_pad_k", True) or K % 16 or K % 64 == 0:
            return False
        if not (x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and 2 <= x.ndim <= 3):
            return False
        if not (2 <= c["data_in_width"] <= 9 and 2 <= c["weight_width"] <= 9) or self.weight_requires_quantisation:
            return False
        xs = [1, K] if x.ndim == 2 else [1, x.shape[-2], K]
        return (ops.resolve_blocking(xs, c["data_in_block_size"], True)[3:] == (1, 16)
                and ops.resolve_blocking([self.out_features, K], c["weight_block_size"], False)[3:] == (1, 16))

    def _forward_block_fp_padded(self, x):
        c, K = self.config, self.in_features
        Kp = (K + 63) // 64 * 64
        x2 = F.pad(x.reshape(-1, K), (0, Kp - K))
        w = self.__dict__.get("_w_bf16_padded")
        if w is None or w[1] != self.weight._version or w[0].device != x.device:
            w = self.__dict__["_w_bf16_padded"] = (ops.bf16_tile(F.pad(self.weight.data, (0, Kp - K))), self.weight._version)
        xt = ops.block_fp_quantize_bf16_tiled(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"])
        y = ops.bf16_gemm_tiled(xt, w[0], x2.shape[0], self.out_features, Kp, self.bias)
        return y.reshape(*x.shape[:-1], self.out_features)

    def _forward_bf16_values(self, x):
        """x (NOT yet quantised) -> x quantiser -> tiled bf16 -> tile GEMM against the tiled quantised weights"""
        c = self.config
        x2 = x.reshape(-1, self.in_features).contiguous()
        if self._w_bf16 is None or self._w_bf16[1] != self.weight._version or self._w_bf16[0].device != x.device:
            self._w_bf16 = (ops.bf16_tile(self.weight.data.contiguous()), self.weight._version)
        xs = [1, self.in_features] if x.ndim == 2 else [1, x.shape[-2], self.in_features]
        if (self.arith == "block_minifloat" and x.ndim <= 3
                and ops.resolve_blocking(xs, c["data_in_block_size"], True)[3:] == (1, 16)):
            # one pass: the block_minifloat values straight into the tiled bf16 operand
            xt = ops.block_minifloat_quantize_bf16_tiled(x2, c["data_in_width"], c["data_in_exponent_width"],
                                                         c["data_in_exponent_bias_width"])
        else:
            # (block_log's all-zero blocks take their value from a reduction over the whole tensor: its own two launches)
            xt = ops.bf16_tile(self.x_quantizer(x).reshape(-1, self.in_features).contiguous())
        y = ops.bf16_gemm_tiled(xt, self._w_bf16[0], x2.shape[0], self.out_features, self.in_features, self.bias)
        return y.reshape(*x.shape[:-1], self.out_features)

    def forward_after(self, x, op, other=None, residual=None):
        """self(relu(x)) (op = "relu": OPT's fc2 behind its activation_fn, modeling_opt.py:412-420) or
        self(silu(x) * other) (op = "silu_mul": Llama's down_proj, modeling_llama.py:216) with the elementwise step read by
        the layer's x quantiser itself -- the reference runs it as torch kernels whose result the quantiser reads back
        (three passes over the [tokens, ffn] tensor instead of one).  Same arithmetic, rounded to fp32 operation by
        operation; whenever the fused quantisers do not apply (first PTQ forward, QAT, bypass, other arithmetics, the
        group flavour) the step runs as torch ops in front of forward().  `residual`: residual + the result (forward_residual)."""
        from ...sharded import ShardedRows, ShardedTiledBf16
        if residual is not None:
            if isinstance(x, (ShardedRows, ShardedTiledBf16)):
                return residual + self.forward_after(x, op, other)
            y = self._forward_after(x, op, other, residual)
            return y
        return self._forward_after(x, op, other, None)

    def _forward_after(self, x, op, other, residual):
        from ...sharded import ShardedRows, ShardedTiledBf16
        if op not in ("relu", "silu_mul") or ((op == "silu_mul") != (other is not None) and not isinstance(x, ShardedTiledBf16)):
            raise ValueError("forward_after: op is 'relu' (no other) or 'silu_mul' (with other)")
        if isinstance(x, ShardedTiledBf16):
            if x.pre_applied != op:
                raise RuntimeError(f"mi355q: a quantised gather that applied {x.pre_applied!r} reached forward_after({op!r})")
            return self._forward_quantised_gather(x)
        if isinstance(x, ShardedRows):
            # fc1's gathered output in the collective's rank-major layout (sharded.shard_model(gather="segments")): the
            # row-aligned route reads the P segments in place with the relu in front; anything else re-assembles it
            if (op == "relu" and not self.bypass and self.is_ptq and not self.weight_requires_quantisation
                    and self._packed_is_current() and self._pending_flavour is None
                    and self._align_mode == "rows" and not self._uses_bf16_route()):
                plan = self._int8_plan(x.buf[0])
                if plan is not None:
                    with torch.no_grad():
                        return self._forward_int8(x, plan, pre=("relu", None))
            x = x.dense()
        # (a layer packed when its weights arrived still carries both flavours: its first forward must see the POST-op
        #  activations to settle the route and drop the other one -- ADVICE r3 -- so it goes through forward() once)
        fused = (self.arith == "block_fp" and self.is_ptq and not self.bypass and not self.weight_requires_quantisation
                 and getattr(self, "_pending_flavour", None) is None
                 and x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and (x.requires_grad or (
                     other is not None and other.requires_grad)))
                 and (other is None or (other.shape == x.shape and other.dtype == x.dtype and other.device == x.device)))
        if fused:
            plan = self._int8_plan(x)
            fused = (plan is not None and self._packed_is_current()
                     and (self._uses_bf16_route() or self._align_mode == "rows"))
        if fused:
            with torch.no_grad():
                o2 = None if other is None else other.reshape(-1, self.in_features)
                if residual is not None and self._residual_fits(x, residual):
                    return self._forward_int8(x, plan, pre=(op, o2), residual=residual)
                y = self._forward_int8(x, plan, pre=(op, o2))
            return y if residual is None else residual + y
        y = self(F.relu(x) if op == "relu" else F.silu(x) * other)
        return y if residual is None else residual + y

    def _bf16_weight_operand(self, device):
        """the quantised weights as the tiled bf16 operand of the per-block-exponent route"""
        if self._w_packed is not None and not self._w_packed.row_scale_flavour:
            return self._w_packed.expand()                   # width-bit storage -> scratch tiled bf16
        if self._w_bf16 is None or self._w_bf16[1] != self.weight._version or self._w_bf16[0].device != device:
            if self._fp32_released:
                raise RuntimeError("mi355q: the fp32 weights were released; this layer cannot switch routes any more")
            self._w_bf16 = (ops.bf16_tile(self.weight.data), self.weight._version)
        return self._w_bf16[0]

    def _residual_fits(self, x, residual) -> bool:
        """can `residual + self(x)` run as one launch?  (the per-block-exponent route's product adds it in its stores)"""
        return (residual is not None and self.is_ptq and not self.bypass and not self.weight_requires_quantisation
                and self._pending_flavour is None and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32
                and self._packed_is_current() and self._int8_plan(x) is not None
                and (self._uses_bf16_route() or self._residual_rides_the_int8_product())
                and residual.dtype == torch.float32 and residual.device == x.device
                and tuple(residual.shape) == tuple(x.shape[:-1]) + (self.out_features,) and residual.is_contiguous()
                and self.out_features % 4 == 0 and residual.data_ptr() % 16 == 0
                and not (torch.is_grad_enabled() and (x.requires_grad or residual.requires_grad)))

    def _residual_rides_the_int8_product(self) -> bool:
        """the row-scale int8 route in its one-launch form (ops.bfp_gemm_aligned(residual=...), round 6): 120-entry activation buckets,
        K a multiple of 128, not the mixed contraction; dense input (a ShardedRows input keeps the separate add)"""
        return (self._align_mode == "rows" and self._mixed is None and self._x_cap == ops.ROW_BUCKET_CAP and self.in_features % 128 == 0
                and self._packed is not None)

    def _residual_operand_ok(self, residual, lead) -> bool:
        """`residual` can ride in the bf16 product's stores (mi355q_bf16_gemm_tiled_res): fp32, contiguous, [.., out_features]"""
        return (torch.is_tensor(residual) and residual.is_cuda and residual.dtype == torch.float32 and residual.is_contiguous()
                and tuple(residual.shape) == tuple(lead) + (self.out_features,) and self.out_features % 4 == 0
                and residual.data_ptr() % 16 == 0 and not (torch.is_grad_enabled() and residual.requires_grad))

    def forward_residual(self, x, residual):
        """residual + self(x) (modeling_llama.py:259, modeling_opt.py:375, 425: the add a decoder layer puts behind o_proj / fc2) --
        in the product's stores where the layer runs on the per-block-exponent route (same bits), as two steps otherwise"""
        if self._residual_fits(x, residual):
            with torch.no_grad():
                return self._forward_int8(x, self._int8_plan(x), residual=residual)
        return residual + self(x)

    def accepts_tiled_input(self) -> bool:
        """can a producer hand this layer its quantised activations as a tiled bf16 operand (`forward_tiled`)?  block_fp PTQ on the
        per-block-exponent route with its weights packed: the attention pass then writes o_proj's operand itself."""
        return (self.arith == "block_fp" and self.is_ptq and not self.bypass and not self.weight_requires_quantisation
                and self._pending_flavour is None and self._packed_is_current() and self._uses_bf16_route()
                and self.config["data_in_width"] <= 9 and self.in_features % 32 == 0
                and ops.resolve_blocking([1, self.in_features], self.config["data_in_block_size"], True)[3:] == (1, 16))

    def consumer_quantiser(self):
        """(width, exponent width, exponent bias) of this layer's activation quantiser, for a producer that applies it"""
        c = self.config
        return (c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"])

    def forward_tiled(self, xt, lead, residual=None):
        """self(x) [+ residual] for x given as `ops.TiledBf16` [rows, in_features] -- this layer's OWN quantised activations, formed by
        the producer (ops.bfp_attention(consumer=self.consumer_quantiser())): the per-block product alone.  `lead`: the leading shape
        of the result (rows = prod(lead))."""
        assert self.accepts_tiled_input() and xt.cols == self.in_features
        with torch.no_grad():
            res2 = residual.reshape(-1, self.out_features) if residual is not None and self._residual_operand_ok(residual, lead) else None
            y = ops.bf16_gemm_tiled(xt.buf, self._bf16_weight_operand(xt.buf.device), xt.rows, self.out_features, self.in_features, self.bias,
                                    out=self._take_out(xt.rows), residual=res2)
        y = y.reshape(*lead, self.out_features)
        return y if residual is None or res2 is not None else residual + y

    def _take_out(self, rows):
        """where the caller wants this call's product stored (sharded.RowShardedLinear: its rank's segment of the all-gather buffer,
        so that the collective runs in place), once; None: a fresh tensor"""
        out, self._out_hint = self.__dict__.get("_out_hint"), None
        if out is not None and out.shape == (rows, self.out_features) and out.dtype == torch.float32 and out.is_contiguous():
            return out
        return None

    def _forward_int8(self, x, plan, pre=None, residual=None):
        x_mbits, w_mbits, xb, wb = plan
        c = self.config
        from ...sharded import ShardedRows
        segments = isinstance(x, ShardedRows)                  # (forward() sends these here on the row-aligned route only)
        x2 = x.buf if segments else x.reshape(-1, self.in_features)
        if self._uses_bf16_route():
            # Activations (or weights) no row window fits: every block keeps its exponent.  A block_fp value of width <= 9
            # is exact in bf16 and a product of two of them exact in fp32, so the product is the bf16 flavour of the tile
            # GEMM (fp32 accumulation, fp32 output): x through the HIP quantiser straight into tiled bf16, the in-place
            # quantised weights tiled once.  |x| <= 1e-8 pass-through elements are rounded to bf16 there (<= 2e-11 each).
            # config["mi355q_blocks_gemm"] = "int8": the blockwise-exact int8 kernel instead (exact integer block dots,
            # several times slower).
            wt = self._bf16_weight_operand(x.device)
            xt = ops.block_fp_quantize_bf16_tiled(x2.contiguous(), c["data_in_width"], c["data_in_exponent_width"],
                                                  c["data_in_exponent_bias"], pre=pre)
            y = ops.bf16_gemm_tiled(xt, wt, x2.shape[0], self.out_features, self.in_features, self.bias, out=self._take_out(x2.shape[0]),
                                    residual=None if residual is None else residual.reshape(-1, self.out_features))
            return y.reshape(*x.shape[:-1], self.out_features)
        if self._mixed is not None and self._mixed["version"] == (self.weight._version, None if self.bias is None else self.bias._version):
            return self._forward_mixed(x, x2, segments, pre, residual)
        assert residual is None or not segments, "a residual behind a segmented input: the caller adds"
        # one fused kernel: quantise + pack + row-align + tile
        xa = ops.block_fp_quantize_aligned_rows(x2, c["data_in_width"], c["data_in_exponent_width"],
                                                c["data_in_exponent_bias"], bucket_cap=self._x_cap, pre=pre,
                                                segments=segments)
        wa = self._w_packed.expand() if self._w_packed is not None else self._packed[0]
        y = ops.bfp_gemm_aligned(xa, wa, self.bias, out=self._take_out(x2.shape[0]),
                                 residual=None if residual is None else residual.reshape(-1, self.out_features))
        if self.align == "auto" and self._x_cap != ops.ROW_NO_ALIGN:
            # results never depend on the mode (an overflowing exception bucket only sends the GEMM to its slow
            # blockwise kernel); look at the overflow word on a doubling schedule and leave row mode if it repeats
            # (not while a HIP graph is being recorded: the read is a host synchronisation)
            self._calls += 0 if ops._capturing() else 1
            if not ops._capturing() and self._calls & (self._calls - 1) == 0 and int(xa.sparse[0]) != 0:
                self._row_overflows += 1
                if self._row_overflows >= 2:
                    self._x_cap = ops.ROW_NO_ALIGN                # activations stopped fitting: no alignment from now on
        return y.reshape(*x.shape[:-1], self.out_features)

    def _forward_mixed(self, x, x2, segments, pre, residual):
        """the mixed contraction: one pass of the class-aware quantiser over x, one launch of the tile kernel (class 0 on the int8
        MFMA, class 1 on the bf16 MFMA); an elementwise step in front, row segments or a residual behind run as torch ops here"""
        c, m = self.config, self._mixed
        if segments:
            x2 = x.dense().reshape(-1, self.in_features)
        if pre is not None:
            x2 = F.relu(x2) if pre[0] == "relu" else F.silu(x2) * pre[1].reshape(-1, self.in_features)
        x0, x1 = ops.block_fp_quantize_classes(x2, m["classes"], c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                               bucket_cap=self._x_cap)
        y = ops.bfp_gemm_mixed(x0, m["wa0"], x1, m["w1"], m["classes"].K1, self.bias, out=self._take_out(x2.shape[0]))
        assert y is not None, "mi355q: the mixed contraction was chosen for a shape the library does not take"
        if self.align == "auto":
            # (as on the row route: an overflowing class-0 bucket only sends the launch to its exact fallback; look at the overflow
            #  word on a doubling schedule of calls and leave for the per-block route if it repeats)
            self._calls += 0 if ops._capturing() else 1
            if not ops._capturing() and self._calls & (self._calls - 1) == 0 and int(x0.sparse[0]) != 0:
                self._row_overflows += 1
                if self._row_overflows >= 2:
                    self._mixed, self._x_cap = None, ops.ROW_NO_ALIGN
        lead = x.shape[:-1]
        y = y.reshape(*lead, self.out_features)
        return y if residual is None else residual + y

    @classmethod
    def from_float(cls, linear_fp32: nn.Linear, config: dict):
        linear = cls(linear_fp32.in_features, linear_fp32.out_features, bias=linear_fp32.bias is not None,
                     config=config)
        with torch.no_grad():
            linear.weight.copy_(linear_fp32.weight)
            if linear.bias is not None:
                linear.bias.copy_(linear_fp32.bias)
        return linear

    def __repr__(self):
        return "{}(in_features={}, out_features={}, bias={}, bypass={}, is_ptq={}, x/w/b-width={}/{}/{})".format(
            self.__class__.__name__, self.in_features, self.out_features, self.bias is not None, self.bypass,
            self.is_ptq, self.config["data_in_width"], self.config["weight_width"],
            self.config.get("bias_width", "NA"))


def gated_mlp(x, gate, up, down, norm=None, residual=None):
    """down(silu(gate(x)) * up(x)) [+ residual] for block_fp PTQ layers (modeling_llama.py:216, the Llama MLP) as TWO launches
    behind the activation quantiser instead of four: x -- with LlamaRMSNorm applied by its quantiser when `norm` = (weight, eps) --
    against gate's and up's weights INTERLEAVED in chunks of 16 rows (ops.interleave_gate_up, built once per pair), whose store
    epilogue forms silu(gate) * up in registers, quantises it with down's activation quantiser and writes down's tiled bf16
    operand (ops.bfp_gemm_aligned_gated: the two [tokens, intermediate] fp32 tensors are never written, the separate
    silu-mul-quantise launch -- 180 MB read, 45 MB written per Llama-7B layer at 2048 tokens -- is gone); then down's product on the
    bf16 flavour of the tile GEMM, the residual in its stores.  Same bits as grouped_linear + down.forward_after.  Returns None
    whenever the pair / the consumer does not qualify (first PTQ forward, gate / up not on the row-scale int8 route, down not on
    the per-block route, shapes, autograd ...): the caller then takes that path."""
    from ...sharded import RowShardedLinear
    layers = (gate, up, down)
    if any(isinstance(l, RowShardedLinear) or not isinstance(l, _LinearBase) for l in layers):
        return None
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and 2 <= x.ndim <= 3) or (torch.is_grad_enabled() and x.requires_grad):
        return None
    if not all(l.arith == "block_fp" and l.is_ptq and not l.bypass and not l.weight_requires_quantisation and l._pending_flavour is None
               and l._packed_is_current() for l in layers):
        return None
    c, dc = gate.config, down.config
    if (c.get("mi355q_fused_gate_up", True) in (False, "off") or not dc.get("mi355q_fused_activation", False) or dc["data_in_width"] > 9
            or gate.in_features != up.in_features or gate.out_features != up.out_features or down.in_features != gate.out_features
            or gate.in_features % 128 or gate.in_features < 256 or gate.out_features % 128 or (gate.bias is None) != (up.bias is None)
            or (norm is not None and len(norm) != 2)):
        return None
    plan = gate._int8_plan(x)
    if plan is None or up._int8_plan(x) != plan or down._int8_plan(x.new_empty((1, down.in_features))) is None:
        return None
    if not all(l._align_mode == "rows" and not l._uses_bf16_route() and l._mixed is None and l._w_packed is None
               and l._x_cap == ops.ROW_BUCKET_CAP for l in (gate, up)):
        return None
    if not all(gate.config[k] == up.config[k] for k in ("data_in_width", "data_in_exponent_width", "data_in_exponent_bias")):
        return None
    if not down._uses_bf16_route() or (down._w_packed is not None and down._w_packed.row_scale_flavour):
        return None
    with torch.no_grad():
        pair = gate.__dict__.get("_gated_pair")
        key = (id(up), gate.weight._version, up.weight._version, gate._packed[0].tiled.data_ptr(), up._packed[0].tiled.data_ptr())
        if pair is None or pair[0] != key:
            w_gu = ops.interleave_gate_up(gate._packed[0], up._packed[0])
            b_gu = None
            if w_gu is not None and gate.bias is not None:
                I = gate.out_features
                b_gu = torch.stack((gate.bias.data.reshape(I // 16, 16), up.bias.data.reshape(I // 16, 16)), dim=1).reshape(-1).contiguous()
            pair = gate.__dict__["_gated_pair"] = (key, w_gu, b_gu)
        _, w_gu, b_gu = pair
        if w_gu is None:
            return None
        x2 = x.reshape(-1, gate.in_features)
        xa = ops.block_fp_quantize_aligned_rows(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                                bucket_cap=gate._x_cap, pre=None if norm is None else ("rmsnorm", norm[0], norm[1]))
        xt = ops.bfp_gemm_aligned_gated(xa, w_gu, dc["data_in_width"], dc["data_in_exponent_width"], dc["data_in_exponent_bias"], b_gu)
        if xt is None:
            return None
        M = x2.shape[0]
        res2 = None
        if residual is not None and down._residual_operand_ok(residual, x.shape[:-1]):
            res2 = residual.reshape(-1, down.out_features)
        y = ops.bf16_gemm_tiled(xt, down._bf16_weight_operand(x.device), M, down.out_features, down.in_features, down.bias,
                                out=down._take_out(M), residual=res2)
        y = y.reshape(*x.shape[:-1], down.out_features)
        return y if residual is None or res2 is not None else residual + y


def relu_mlp(x, fc1, fc2, norm=None, residual=None):
    """fc2(relu(fc1(x))) [+ residual] for block_fp PTQ layers (modeling_opt.py:412-420, the OPT MLP) as TWO launches behind the
    activation quantiser: fc1's product with relu and fc2's activation quantiser in its store epilogue (ops.bfp_gemm_aligned_relu:
    the [tokens, ffn] fp32 tensor is never written, the relu-quantise launch is gone), then fc2's product on the bf16 flavour of the
    tile GEMM.  `norm` = (weight, bias, eps): OPT's final_layer_norm applied by fc1's quantiser.  Same bits as fc1 + fc2.forward_after.
    Returns None whenever the layers do not qualify (the caller then takes that path): gated_mlp's conditions with one producer."""
    from ...sharded import RowShardedLinear
    if any(isinstance(l, RowShardedLinear) or not isinstance(l, _LinearBase) for l in (fc1, fc2)):
        return None
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and 2 <= x.ndim <= 3) or (torch.is_grad_enabled() and x.requires_grad):
        return None
    if not all(l.arith == "block_fp" and l.is_ptq and not l.bypass and not l.weight_requires_quantisation and l._pending_flavour is None
               and l._packed_is_current() for l in (fc1, fc2)):
        return None
    c, dc = fc1.config, fc2.config
    if (c.get("mi355q_fused_gate_up", True) in (False, "off") or not dc.get("mi355q_fused_activation", False) or dc["data_in_width"] > 9
            or fc2.in_features != fc1.out_features or fc1.in_features % 128 or fc1.in_features < 256 or fc1.out_features % 32
            or (norm is not None and len(norm) != 3)):
        return None
    if fc1._int8_plan(x) is None or fc2._int8_plan(x.new_empty((1, fc2.in_features))) is None:
        return None
    if not (fc1._align_mode == "rows" and not fc1._uses_bf16_route() and fc1._mixed is None and fc1._w_packed is None
            and fc1._x_cap == ops.ROW_BUCKET_CAP):
        return None
    if not fc2._uses_bf16_route() or (fc2._w_packed is not None and fc2._w_packed.row_scale_flavour):
        return None
    with torch.no_grad():
        x2 = x.reshape(-1, fc1.in_features)
        xa = ops.block_fp_quantize_aligned_rows(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                                bucket_cap=fc1._x_cap,
                                                pre=None if norm is None else ("layernorm", norm[0], norm[2], norm[1]))
        xt = ops.bfp_gemm_aligned_relu(xa, fc1._packed[0], dc["data_in_width"], dc["data_in_exponent_width"], dc["data_in_exponent_bias"],
                                       fc1.bias)
        if xt is None:
            return None
        M = x2.shape[0]
        res2 = residual.reshape(-1, fc2.out_features) if residual is not None and fc2._residual_operand_ok(residual, x.shape[:-1]) else None
        y = ops.bf16_gemm_tiled(xt, fc2._bf16_weight_operand(x.device), M, fc2.out_features, fc2.in_features, fc2.bias,
                                out=fc2._take_out(M), residual=res2)
        y = y.reshape(*x.shape[:-1], fc2.out_features)
        return y if residual is None or res2 is not None else residual + y


FP32_SPLIT_MIN_ROWS = 128     # (fewer tokens: the product is bound by the weight bytes, and the split operand is 3 x the fp32 one)


def fp32_linear(x, linear: nn.Linear, mode: str = "split"):
    """F.linear(x, linear.weight, linear.bias) for a layer the reference leaves UNQUANTISED -- the language-model head
    (modeling_llama.py:772,866; modeling_opt.py:942-944: nn.Linear in fp32) -- as an fp32-equivalent product on the bf16 MFMA:
    both operands as three bf16 parts, the six part products side by side along K in ONE launch of the bf16 tile GEMM
    (ops.fp32_split_tile / fp32_gemm_split; csrc/mi355q_split.hip).  Closer to an fp64 product than the vendor fp32 GEMM and 1.7 x
    faster at Llama-7B's head.  The weights' operand is built once and kept on the module (rebuilt when the parameter is written).
    `mode` "vendor", gradients wanted, a CPU tensor, fewer than FP32_SPLIT_MIN_ROWS tokens or in_features % 32 != 0: torch's
    F.linear, counted as a vendor GEMM."""
    w = linear.weight
    M = x.numel() // max(1, x.shape[-1])
    ok = (mode == "split" and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.device == x.device
          and linear.in_features % 32 == 0 and M >= FP32_SPLIT_MIN_ROWS and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)))
    if not ok:
        ops.count_vendor_gemm("fp32_linear (unquantised layer, vendor fp32 GEMM)")
        return F.linear(x, w, linear.bias)
    try:
        key = (w.data_ptr(), w._version, str(w.device))
    except RuntimeError:                      # (inference-mode tensors have no version counter: keyed by storage alone)
        key = (w.data_ptr(), None, str(w.device))
    cached = linear.__dict__.get("_mi355q_split_weight")
    if cached is None or cached[0] != key:
        with torch.no_grad():
            cached = (key, ops.fp32_split_tile(w.detach().contiguous(), 1))
        linear.__dict__["_mi355q_split_weight"] = cached
    with torch.no_grad():
        x2 = x.reshape(-1, linear.in_features).contiguous()
        y = ops.fp32_gemm_split(ops.fp32_split_tile(x2, 0), cached[1], M, linear.out_features, linear.in_features, bias=linear.bias)
    return y.reshape(*x.shape[:-1], linear.out_features)


def grouped_linear(x, layers, norm=None):
    """[layer(x) for layer in layers] for block_fp PTQ Linear layers that take the SAME input and have the same shape and
    widths -- the q / k / v projections of an attention block, gate / up of a gated MLP, which the reference's modules
    call one after the other (modeling_opt.py:231-245, modeling_llama.py:216, 283-287) -- as ONE activation quantisation
    and ONE launch of the tile GEMM over all their column tiles (ops.bfp_gemm_aligned_multi): the separate products
    leave compute units idle (2048 -> 2048: 128 tiles each) or waste most of a second round (4096 -> 11008: 344 tiles).
    Bit-identical to the separate calls; falls back to them whenever the group does not qualify (first PTQ forward,
    other arithmetics, the per-block bf16 route, differing shapes ...).

    `norm` = (weight, eps): the layers take LlamaRMSNorm(x) (modeling_llama.py:81-92, 236-238: the input of q / k / v and of
    gate / up, which nothing else reads), `norm` = (weight, bias, eps): nn.LayerNorm(x) (OPT's self_attn_layer_norm in
    front of q / k / v and final_layer_norm in front of fc1, modeling_opt.py:391-415) -- and the quantiser, which holds a
    whole row per workgroup, applies the norm itself, so the normalised tensor is never written.  Then the mean of squares is summed in the kernel's own fixed
    order: results agree with the separate norm to within the last-bit differences any two fp32 summation orders
    show (torch's own CPU and GPU reductions included), not bit for bit."""
    layers = list(layers)
    from ...sharded import RowShardedLinear
    if all(isinstance(l, RowShardedLinear) for l in layers):
        if all(l.keep_local for l in layers):
            # head-sharded q / k / v (sharded.shard_model(heads=True)): the rank's own heads, no collective here
            return grouped_linear(x, [l.local for l in layers], norm=norm)
        # row-sharded projections (sharded.shard_model): this rank's shards as one group, one all-gather per projection
        for l in layers:                                       # (each product straight into its rank's segment of its gather buffer)
            l._aim_at_gather_buffer(x)
        try:
            ys = grouped_linear(x, [l.local for l in layers], norm=norm)
        finally:
            for l in layers:
                l.local._out_hint = None
        if len(layers) == 2 and layers[0].gather == "quantised" and layers[0].consumer_pre == "silu_mul":
            # Llama's gate / up in front of down_proj (sharded.shard_model(gather="quantised")): both shards of a rank cover the
            # same columns, so silu(gate) * up and down_proj's quantiser run on the rank's own slice; ONE all-gather, of the
            # tiled bf16 operand.  The second result is None: down_proj.forward_after(gate, "silu_mul", None) reads the first
            return [layers[0].gather_output(ys[0], other=ys[1]), None]
        return [l.gather_output(y) for l, y in zip(layers, ys)]
    first = layers[0]

    def normed():
        if len(norm) == 3:
            return F.layer_norm(x, (x.shape[-1],), norm[0], norm[1], norm[2])
        w, eps = norm
        v = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(v + eps)).to(x.dtype)
    ok = (len(layers) in ((1, 2, 3) if norm is not None else (2, 3)) and all(isinstance(l, _LinearBase) and l.arith == "block_fp" and l.is_ptq and not l.bypass
                                         and not l.weight_requires_quantisation for l in layers)
          and not (torch.is_grad_enabled() and x.requires_grad))
    if ok:
        plan = first._int8_plan(x)
        ok = plan is not None and (norm is None or (x.is_cuda and x.dtype == torch.float32)) and all(
            l._packed_is_current() and l._align_mode == "rows" and not l._uses_bf16_route() and l._mixed is None
            and (l._w_packed is None or (l._w_packed.row_scale_flavour and l._pending_flavour is None))
            and l.in_features == first.in_features and l.out_features == first.out_features and l._x_cap == first._x_cap
            and l._x_cap == ops.ROW_BUCKET_CAP and l._int8_plan(x) == plan
            and all(l.config[k] == first.config[k] for k in ("data_in_width", "data_in_exponent_width", "data_in_exponent_bias"))
            for l in layers)
    if ok:
        c = first.config
        x2 = x.reshape(-1, first.in_features)
        with torch.no_grad():
            xa = ops.block_fp_quantize_aligned_rows(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                                    bucket_cap=first._x_cap,
                                                    pre=None if norm is None else (
                                                        ("rmsnorm", norm[0], norm[1]) if len(norm) == 2 else
                                                        ("layernorm", norm[0], norm[2], norm[1])))
            # one launch, or the split of the group that takes fewer rounds over the chip (ops.grouped_launch_plan)
            # (width-bit storage: a launch's members expand into scratch slots 0 .. g - 1 first -- round 5; before, a packed layer
            #  kept its group off this path: separate launches, the norm by six torch kernels)
            outs, at = [], 0
            for g in ops.grouped_launch_plan(x2.shape[0], first.out_features, len(layers)):
                part = layers[at:at + g]
                at += g
                was = [l._w_packed.expand(i) if l._w_packed is not None else l._packed[0] for i, l in enumerate(part)]
                hints = [l._take_out(x2.shape[0]) for l in part]
                if g == 1:
                    ys = [ops.bfp_gemm_aligned(xa, was[0], part[0].bias, out=hints[0])]
                else:
                    ys = ops.bfp_gemm_aligned_multi(xa, was, [l.bias for l in part], outs=hints)
                if ys is None:
                    outs = None
                    break
                outs.extend(ys)
        if outs is not None:
            return [y.reshape(*x.shape[:-1], first.out_features) for y in outs]
    # The per-block-exponent route (bf16 tile GEMM: inputs no row window fits -- every Linear of a model whose hidden channels
    # differ in magnitude): ONE activation operand for the group, LlamaRMSNorm applied by its quantiser; the products stay separate
    # launches.  (Before round 5 such a group fell back to the torch norm -- six elementwise kernels -- and quantised x once per layer.)
    if (len(layers) >= 1 and (norm is None or len(norm) == 2) and x.is_cuda and x.dtype == torch.float32 and 2 <= x.ndim <= 3
            and not (torch.is_grad_enabled() and x.requires_grad)
            and all(isinstance(l, _LinearBase) and l.arith == "block_fp" and l.is_ptq and not l.bypass and not l.weight_requires_quantisation
                    and l._pending_flavour is None for l in layers)):
        plan = first._int8_plan(x)
        if (plan is not None and all(l._packed_is_current() and l._uses_bf16_route() and l.in_features == first.in_features
                                     and (l._w_packed is None or not l._w_packed.row_scale_flavour or len(layers) == 1)
                                     and l._int8_plan(x) == plan
                                     and all(l.config[k] == first.config[k] for k in ("data_in_width", "data_in_exponent_width", "data_in_exponent_bias"))
                                     for l in layers)
                and (norm is not None or len(layers) > 1)):
            c = first.config
            x2 = x.reshape(-1, first.in_features).contiguous()
            with torch.no_grad():
                xt = ops.block_fp_quantize_bf16_tiled(x2, c["data_in_width"], c["data_in_exponent_width"], c["data_in_exponent_bias"],
                                                      pre=None if norm is None else ("rmsnorm", norm[0], norm[1]))
                outs = []
                for l in layers:           # (a packed layer's expand() shares one scratch operand: each product before the next expand)
                    y = ops.bf16_gemm_tiled(xt, l._bf16_weight_operand(x.device), x2.shape[0], l.out_features, l.in_features, l.bias,
                                            out=l._take_out(x2.shape[0]))
                    outs.append(y.reshape(*x.shape[:-1], l.out_features))
            return outs
    h = x if norm is None else normed()
    return [l(h) for l in layers]


def _linear_class(name: str, arith: str):
    return type(name, (_LinearBase,), {"arith": arith, "__module__": __name__,
                                       "__doc__": f"nn.Linear with {arith} input / weight / bias quantisers"})


LinearBlockFP = _linear_class("LinearBlockFP", "block_fp")
LinearBlockMinifloat = _linear_class("LinearBlockMinifloat", "block_minifloat")
LinearBlockLog = _linear_class("LinearBlockLog", "block_log")
LinearInteger = _linear_class("LinearInteger", "integer")
LinearLog = _linear_class("LinearLog", "log")
LinearMinifloatDenorm = _linear_class("LinearMinifloatDenorm", "minifloat_denorm")
LinearMinifloatIEEE = _linear_class("LinearMinifloatIEEE", "minifloat_ieee")
