#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference in the build container.

Run here only (needs /root/reference); the GPU box never sees the reference.
Writes small .npz / .json fixtures into tests/golden/.  Fixtures are data: seeded
inputs and the reference's outputs (plus integer internals tapped from the
reference's own torch ops with a TorchDispatchMode).

    python tools/gen_golden.py            # regenerate everything
"""
from __future__ import annotations

import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/src/llm_mixed_q/models/quantize")
OUT = ROOT / "tests" / "golden"
CFG_DIR = Path("/root/reference/experiments/emnlp/configs")


def _load_ref():
    """Load reference `models/quantize` as package `refq` (needs only torch+numpy; optuna is
    referenced by the sampler module only, so a dummy module satisfies the import)."""
    if "optuna" not in sys.modules:
        opt = types.ModuleType("optuna")
        opt.Trial = object
        opt.Study = object
        opt.trial = types.SimpleNamespace(FrozenTrial=object)
        sys.modules["optuna"] = opt
        sys.modules["optuna.trial"] = types.ModuleType("optuna.trial")
        sys.modules["optuna.trial"].FrozenTrial = object
    spec = importlib.util.spec_from_file_location(
        "refq", REF / "__init__.py", submodule_search_locations=[str(REF)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refq"] = mod
    spec.loader.exec_module(mod)
    return mod


class Tap(TorchDispatchMode):
    """Record outputs of the aten ops that carry the integers of the formats."""
    WATCH = ("sign", "ceil", "floor", "round", "clamp", "log2")

    def __init__(self):
        super().__init__()
        self.log = []

    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in self.WATCH and isinstance(out, torch.Tensor):
            self.log.append((name, out.detach().clone()))
        return out


def _canon(t: torch.Tensor, kind: str) -> np.ndarray:
    """reference blocked layouts -> [n_blocks, block_elems] / [n_blocks]."""
    a = t.numpy()
    if kind == "bias1d" or kind == "act2d":       # [nb, B] / [N, nb, B]  (max: [..., 1])
        return a.reshape(-1, a.shape[-1])
    if kind == "weight2d":                        # [be, L]  (max: [1, L])
        return np.ascontiguousarray(a.T)
    if kind == "act3d":                           # [B, be, L]  (max: [B, 1, L])
        return np.ascontiguousarray(a.transpose(0, 2, 1)).reshape(-1, a.shape[1])
    raise KeyError(kind)


def _kind(x: torch.Tensor, skip_first_dim: bool) -> str:
    return {1: "bias1d", 2: "act2d" if skip_first_dim else "weight2d", 3: "act3d"}[x.ndim]


def rand_tensor(shape, seed, style="randn"):
    g = torch.Generator().manual_seed(seed)
    if style == "randn":
        return torch.randn(*shape, generator=g)
    if style == "rowscale":     # activations with a wide per-row scale spread
        x = torch.randn(*shape, generator=g)
        s = torch.exp(2.0 * torch.randn(*shape[:-1], 1, generator=g))
        return x * s
    if style == "weight":
        return torch.randn(*shape, generator=g) * 0.02
    if style == "outlier":      # a few huge channels, the LLM activation pattern
        x = torch.randn(*shape, generator=g)
        idx = torch.randint(0, shape[-1], (max(1, shape[-1] // 64),), generator=g)
        x[..., idx] *= 60.0
        return x
    if style == "sparse":       # zeros, zero blocks, tiny values
        x = torch.randn(*shape, generator=g)
        m = torch.rand(*shape, generator=g)
        x = torch.where(m < 0.35, torch.zeros_like(x), x)
        x = torch.where((m > 0.35) & (m < 0.45), x * 1e-9, x)
        flat = x.reshape(-1)
        n = flat.numel()
        for s in range(0, n - 48, 211):      # whole zero stretches -> all-zero blocks
            flat[s:s + 48] = 0
        return flat.reshape(shape)
    if style == "big":
        return torch.randn(*shape, generator=g) * 300.0
    raise KeyError(style)


def edge_values() -> np.ndarray:
    """H1 boundary set: 2**k(1+j 2**-23), values just below 2**k, around 2**(k+1/2), 1e-8/1e-9
    neighbourhoods, exact powers of two, half-way mantissas."""
    out = []
    ks = np.concatenate([np.arange(-34, 35, 3), np.arange(-6, 7)])
    ks = np.unique(ks)
    js = np.arange(0, 8)
    bits = ((ks[:, None] + 127).astype(np.uint32) << 23) | js[None, :].astype(np.uint32)
    out.append(bits.view(np.float32).reshape(-1))
    bits = ((ks[:, None] + 127).astype(np.uint32) << 23) | (0x7FFFFF - js[None, :]).astype(np.uint32)
    out.append(bits.view(np.float32).reshape(-1))
    m0 = int((2 ** 0.5 - 1) * 2 ** 23)
    bits = ((ks[:, None] + 127).astype(np.uint32) << 23) | (m0 + np.arange(-6, 7))[None, :].astype(np.uint32)
    out.append(bits.view(np.float32).reshape(-1))
    near = np.float32(1e-8)
    out.append(np.array([np.nextafter(near, np.float32(0)), near, np.nextafter(near, np.float32(1)),
                         1e-9, -1e-9, 2e-9, 3e-9, -3e-9, 9e-9, 1.1e-8, -1.1e-8, 0.0, -0.0,
                         5e-10, -5e-10, 1e-10, 1e-7, -1e-7], dtype=np.float32))
    # half-way mantissas at several widths: (i + 0.5) / 2**mb for block max 1 -> exponent 0
    for mb in (1, 3, 5, 7):
        i = np.arange(0, 2 ** mb)
        out.append(((i + 0.5) / 2 ** mb).astype(np.float32))
        out.append((-(i + 0.5) / 2 ** mb).astype(np.float32))
    v = np.concatenate(out)
    v = np.concatenate([v, -v[: v.size // 3]])
    return v.astype(np.float32)


def edge_tensors():
    v = edge_values()
    r = np.random.default_rng(7)
    # (a) each edge value as the max of its own block (rest smaller) -> exercises block exponent
    n = v.size
    a = (r.uniform(-0.45, 0.45, size=(n, 16)).astype(np.float32)) * np.abs(v)[:, None]
    a[:, 5] = v
    # (b) edge values packed densely -> exercises per-element decisions
    pad = (-n) % 16
    b = np.concatenate([v, np.zeros(pad, np.float32)]).reshape(-1, 16)
    r.shuffle(b, axis=0)
    # (c) zero blocks / mixed / all-zero tensor
    c = np.zeros((6, 64), np.float32)
    c[1, 3] = 0.75
    c[2, 16:32] = r.normal(size=16).astype(np.float32) * 1e-3
    c[4, :] = r.normal(size=64).astype(np.float32)
    c[4, 32:48] = 0
    d = np.zeros((3, 32), np.float32)
    return {"blockmax": a, "dense": b, "zeros": c, "allzero": d}


BFP = lambda w, bs, bias=127, ew=8: dict(width=w, exponent_width=ew, exponent_bias=bias, block_size=bs)
BM = lambda bs, w=8, ew=4, ebw=8: dict(width=w, exponent_width=ew, exponent_bias_width=ebw, block_size=bs)
BL = lambda bs, w=8, ebw=8: dict(width=w, exponent_bias_width=ebw, block_size=bs)


def run_case(refq, qname, x, params, skip):
    fn = refq.quantizers.QUANTIZER_MAP[qname]
    with Tap() as tap:
        y = fn(x.clone(), **params, skip_first_dim=skip)
    kind = _kind(x, skip)
    log = tap.log
    rec = {"x": x.numpy(), "y": y.numpy()}
    names = [n for n, _ in log]
    if qname == "block_fp":
        # sign[elt] ceil[blk] clamp[blk] round[elt] clamp[elt]
        i_ceil = names.index("ceil")
        sign = log[names.index("sign")][1]
        exp = log[i_ceil + 1][1]
        assert log[i_ceil + 1][0] == "clamp"
        i_round = names.index("round")
        mant = log[i_round + 1][1]
        assert log[i_round + 1][0] == "clamp"
        rec["exp"] = _canon(exp, kind).reshape(-1).astype(np.int16)
        rec["mant"] = (_canon(sign, kind) * _canon(mant, kind)).astype(np.int8)
    elif qname == "block_minifloat":
        # log2,floor,clamp[blk]=bias ; sign ; log2,floor,clamp[elt]=exp ; round,clamp (normal) ; round,clamp (subnormal)
        clamps = [t for n, t in log if n == "clamp"]
        sign = log[names.index("sign")][1]
        bias, exp, m_n, m_s = clamps[0], clamps[1], clamps[2], clamps[3]
        normal = exp != -bias
        mant = torch.where(normal, m_n, m_s)
        rec["bias"] = _canon(bias, kind).reshape(-1).astype(np.int16)
        rec["sign"] = _canon(sign, kind).astype(np.int8)
        rec["exp"] = _canon(exp, kind).astype(np.int16)
        rec["mant"] = _canon(mant, kind).astype(np.int16)
    elif qname == "block_log":
        clamps = [t for n, t in log if n == "clamp"]
        sign = log[names.index("sign")][1]
        bias, exp = clamps[0], clamps[1]
        rec["bias"] = _canon(bias, kind).reshape(-1).astype(np.int16)
        rec["sign"] = _canon(sign, kind).astype(np.int8)
        rec["exp"] = _canon(exp, kind).astype(np.int16)
    return rec


def gen_quantizers(refq):
    cases = {}

    def add(tag, qname, x, params, skip):
        rec = run_case(refq, qname, x, params, skip)
        meta = dict(quantizer=qname, params=params, skip_first_dim=skip, shape=list(x.shape))
        cases[tag] = meta
        for k, v in rec.items():
            arrays[f"{tag}/{k}"] = v

    arrays = {}
    n = 0
    # G1/G2: block_fp, widths 2..8, four layouts, ragged shapes
    for w in (2, 3, 4, 5, 6, 7, 8):
        for style in ("randn", "rowscale", "outlier", "sparse"):
            add(f"bfp_w{w}_act2d_{style}", "block_fp", rand_tensor((12, 80), 100 + n, style), BFP(w, [1, 16]), True); n += 1
        add(f"bfp_w{w}_act2d_ragged", "block_fp", rand_tensor((5, 43), 100 + n, "rowscale"), BFP(w, [1, 16]), True); n += 1
        add(f"bfp_w{w}_act3d", "block_fp", rand_tensor((3, 7, 48), 100 + n, "rowscale"), BFP(w, [1, 16]), True); n += 1
        add(f"bfp_w{w}_weight", "block_fp", rand_tensor((24, 64), 100 + n, "weight"), BFP(w, [1, 16]), False); n += 1
        add(f"bfp_w{w}_weight_ragged", "block_fp", rand_tensor((10, 37), 100 + n, "weight"), BFP(w, [1, 16]), False); n += 1
        add(f"bfp_w{w}_bias", "block_fp", rand_tensor((50,), 100 + n, "weight"), BFP(w, [16]), False); n += 1
    # exponent_bias None, narrow exponent widths (clamping), other block shapes
    add("bfp_w6_biasNone", "block_fp", rand_tensor((8, 64), 900, "rowscale"), BFP(6, [1, 16], None), True)
    add("bfp_w6_ew4", "block_fp", rand_tensor((8, 64), 901, "big"), BFP(6, [1, 16], None, 4), True)
    add("bfp_w6_ew3_small", "block_fp", rand_tensor((8, 64), 902, "weight"), BFP(6, [1, 16], 3, 3), True)
    add("bfp_w8_blk32", "block_fp", rand_tensor((6, 96), 903, "randn"), BFP(8, [1, 32]), True)
    add("bfp_w6_blk8", "block_fp", rand_tensor((6, 40), 904, "randn"), BFP(6, [1, 8]), True)
    add("bfp_w6_blk1d_on2d", "block_fp", rand_tensor((6, 40), 905, "randn"), BFP(6, [16]), True)
    add("bfp_w6_weight_2dblk", "block_fp", rand_tensor((10, 24), 906, "weight"), BFP(6, [4, 8]), False)
    add("bfp_w6_weight_1entry", "block_fp", rand_tensor((6, 40), 907, "weight"), BFP(6, [16]), False)
    add("bfp_w6_act3d_2dblk", "block_fp", rand_tensor((2, 6, 40), 908, "randn"), BFP(6, [2, 16]), True)
    add("bfp_w6_act3d_1entry", "block_fp", rand_tensor((2, 5, 40), 909, "randn"), BFP(6, [16]), True)
    add("bfp_w6_blk_gt_dim", "block_fp", rand_tensor((4, 10), 910, "randn"), BFP(6, [1, 16]), True)
    # block_minifloat / block_log, shipped params and a few others
    for style in ("randn", "rowscale", "outlier", "sparse", "big", "weight"):
        add(f"bm_act2d_{style}", "block_minifloat", rand_tensor((12, 80), 300 + n, style), BM([1, 16]), True); n += 1
        add(f"bl_act2d_{style}", "block_log", rand_tensor((12, 80), 300 + n, style), BL([1, 16]), True); n += 1
    for qn, P in (("block_minifloat", BM), ("block_log", BL)):
        s = "bm" if qn == "block_minifloat" else "bl"
        add(f"{s}_act2d_ragged", qn, rand_tensor((5, 43), 400 + n, "rowscale"), P([1, 16]), True); n += 1
        add(f"{s}_act3d", qn, rand_tensor((3, 7, 48), 400 + n, "big"), P([1, 16]), True); n += 1
        add(f"{s}_weight", qn, rand_tensor((24, 64), 400 + n, "randn"), P([1, 16]), False); n += 1
        add(f"{s}_weight_ragged", qn, rand_tensor((10, 37), 400 + n, "big"), P([1, 16]), False); n += 1
        add(f"{s}_bias", qn, rand_tensor((50,), 400 + n, "randn"), P([16]), False); n += 1
    add("bm_w6_e3", "block_minifloat", rand_tensor((8, 64), 950, "big"), BM([1, 16], 6, 3, 4), True)
    add("bm_w8_e5_ebw3", "block_minifloat", rand_tensor((8, 64), 951, "big"), BM([1, 16], 8, 5, 3), True)
    add("bl_w4", "block_log", rand_tensor((8, 64), 952, "rowscale"), BL([1, 16], 4, 8), True)
    add("bl_w6_ebw4", "block_log", rand_tensor((8, 64), 953, "big"), BL([1, 16], 6, 4), True)
    add("bl_w8_weight_small", "block_log", rand_tensor((8, 64), 954, "weight"), BL([1, 16]), False)
    # G3 edge tensors through every format
    for name, arr in edge_tensors().items():
        x = torch.from_numpy(arr.copy())
        for w in (4, 6, 8):
            add(f"edge_{name}_bfp_w{w}", "block_fp", x, BFP(w, [1, 16]), True)
        add(f"edge_{name}_bfp_w6_weight", "block_fp", x, BFP(6, [1, 16]), False)
        add(f"edge_{name}_bm", "block_minifloat", x, BM([1, 16]), True)
        add(f"edge_{name}_bl", "block_log", x, BL([1, 16]), True)
        # scale the edge set so per-element decisions of minifloat/log are not flushed
        xs = x * 64.0
        add(f"edge_{name}x64_bm", "block_minifloat", xs, BM([1, 16]), True)
        add(f"edge_{name}x64_bl", "block_log", xs, BL([1, 16]), True)
    np.savez_compressed(OUT / "quantizers.npz", **arrays)
    (OUT / "quantizers.json").write_text(json.dumps(cases, indent=1))
    print(f"quantizers: {len(cases)} cases, {sum(a.nbytes for a in arrays.values())/1e6:.2f} MB raw")


def load_toml(path):
    import tomli
    cfg = tomli.loads(Path(path).read_text())

    def fix(d):
        for k, v in list(d.items()):
            if isinstance(v, dict):
                fix(v)
            elif v == "NA":
                d[k] = None
        return d
    return fix(cfg)


def gen_modules(refq):
    """G4: PTQ Linear forward (first and second call) and matmul/bmm, per format."""
    arrays, cases = {}, {}
    tomls = {"bfp_6bit": "bfp_6bit.toml", "bfp_4bit": "bfp_4bit.toml", "block_fp": "block_fp.toml",
             "block_minifloat": "block_minifloat.toml", "block_log": "block_log.toml"}
    for tag, fname in tomls.items():
        cfg = load_toml(CFG_DIR / "quantization" / fname)["default"]
        lin_cfg = refq.parse_node_config(dict(cfg), "linear")
        mm_cfg = refq.parse_node_config(dict(cfg), "matmul")
        cases[tag] = {"linear_config": lin_cfg, "matmul_config": mm_cfg}
        scale = 40.0 if tag in ("block_minifloat",) else 1.0
        for has_bias in (True, False):
            torch.manual_seed(11)
            cls = refq.get_quantized_cls("linear", lin_cfg)
            fp = torch.nn.Linear(96, 48, bias=has_bias)
            with torch.no_grad():
                fp.weight.mul_(scale * 3)
            lin = cls.from_float(fp, lin_cfg)
            x1 = rand_tensor((2, 9, 96), 21, "rowscale") * scale
            x2 = rand_tensor((14, 96), 22, "outlier") * scale
            w0 = lin.weight.detach().clone()
            b0 = lin.bias.detach().clone() if has_bias else None
            y1 = lin(x1)
            wq = lin.weight.detach().clone()
            y2 = lin(x2)
            k = f"{tag}/linear_bias{int(has_bias)}"
            arrays[f"{k}/w"] = w0.numpy()
            if has_bias:
                arrays[f"{k}/b"] = b0.numpy()
                arrays[f"{k}/bq"] = lin.bias.detach().numpy().copy()
            arrays[f"{k}/x1"], arrays[f"{k}/y1"] = x1.numpy(), y1.detach().numpy()
            arrays[f"{k}/x2"], arrays[f"{k}/y2"] = x2.numpy(), y2.detach().numpy()
            arrays[f"{k}/wq"] = wq.numpy()
        # attention-shaped matmul / bmm: [h,T,d]x[h,d,T], [h,T,T]x[h,T,d], 4-D matmul, 2-D matmul
        q = rand_tensor((3, 20, 32), 31, "rowscale") * scale
        kt = rand_tensor((3, 32, 20), 32, "randn") * scale
        p = torch.softmax(rand_tensor((3, 20, 20), 33, "randn"), -1) * (scale if scale > 1 else 1)
        v = rand_tensor((3, 20, 32), 34, "randn") * scale
        bmm = refq.get_quantized_func("bmm", mm_cfg)
        mm = refq.get_quantized_func("matmul", mm_cfg)
        arrays[f"{tag}/bmm0/x"], arrays[f"{tag}/bmm0/y"], arrays[f"{tag}/bmm0/out"] = q.numpy(), kt.numpy(), bmm(q, kt, mm_cfg).numpy()
        arrays[f"{tag}/bmm1/x"], arrays[f"{tag}/bmm1/y"], arrays[f"{tag}/bmm1/out"] = p.numpy(), v.numpy(), bmm(p, v, mm_cfg).numpy()
        q4, k4 = q.reshape(1, 3, 20, 32), kt.reshape(1, 3, 32, 20)
        arrays[f"{tag}/mm4d/x"], arrays[f"{tag}/mm4d/y"], arrays[f"{tag}/mm4d/out"] = q4.numpy(), k4.numpy(), mm(q4, k4, mm_cfg).numpy()
        a2, b2 = q[0], kt[0]
        arrays[f"{tag}/mm2d/x"], arrays[f"{tag}/mm2d/y"], arrays[f"{tag}/mm2d/out"] = a2.numpy(), b2.numpy(), mm(a2, b2, mm_cfg).numpy()
    np.savez_compressed(OUT / "modules.npz", **arrays)
    (OUT / "modules.json").write_text(json.dumps(cases, indent=1))
    print(f"modules: {len(arrays)} arrays, {sum(a.nbytes for a in arrays.values())/1e6:.2f} MB raw")


def gen_config_and_profile(refq):
    """G6: parse_node_config for every shipped quantisation TOML x op, and the analytic
    layer profiler for representative shapes."""
    out = {"parse_node_config": {}, "profile_linear_layer": [], "profile_matmul_layer": []}
    for f in sorted((CFG_DIR / "quantization").glob("*.toml")):
        cfg = load_toml(f)
        for section, body in cfg.items():
            if not isinstance(body, dict) or "name" not in body:
                continue
            for op in ("linear", "matmul", "bmm", "rotary_positional_encoding"):
                try:
                    parsed = refq.parse_node_config(dict(body), op, strict=True)
                    out["parse_node_config"][f"{f.name}:{section}:{op}"] = {"in": body, "out": parsed}
                except Exception as e:  # the reference raises KeyError for missing entries
                    out["parse_node_config"][f"{f.name}:{section}:{op}"] = {"in": body, "raises": type(e).__name__}
    bfp6 = load_toml(CFG_DIR / "quantization" / "bfp_6bit.toml")["default"]
    integer = load_toml(CFG_DIR / "quantization" / "integer.toml")["default"]
    for cfg in (bfp6, integer, dict(bfp6, bypass=True)):
        for (i, o, b, n) in ((768, 3072, True, 2048), (4096, 11008, False, 2048), (100, 37, True, 5)):
            try:
                r = refq.profile_linear_layer(cfg, i, o, b, n)
                out["profile_linear_layer"].append({"cfg": cfg, "args": [i, o, b, n], "out": {k: int(v) for k, v in r.items()}})
            except Exception as e:
                out["profile_linear_layer"].append({"cfg": cfg, "args": [i, o, b, n], "raises": type(e).__name__})
        for (s0, s1) in (((2048, 64), (64, 2048)), ((2048, 2048), (2048, 128)), ((7, 33), (33, 5))):
            try:
                r = refq.profile_matmul_layer(cfg, s0, s1)
                out["profile_matmul_layer"].append({"cfg": cfg, "args": [list(s0), list(s1)], "out": {k: int(v) for k, v in r.items()}})
            except Exception as e:
                out["profile_matmul_layer"].append({"cfg": cfg, "args": [list(s0), list(s1)], "raises": type(e).__name__})
    (OUT / "config_profile.json").write_text(json.dumps(out, indent=1))
    print(f"config/profile: {len(out['parse_node_config'])} parse cases")


def gen_elementwise(refq):
    """The un-blocked quantisers (minifloat.py:21-196, log.py:22-56): reference outputs on the edge sets and random
    tensors, several (width, exponent width, bias) settings each -> tests/golden/elementwise.{npz,json}"""
    r = np.random.default_rng(11)
    inputs = {"edge": edge_values().reshape(-1), "randn": r.normal(size=(37, 53)).astype(np.float32),
              "scaled": (r.normal(size=(16, 64)) * np.exp(r.normal(size=(16, 1)) * 4)).astype(np.float32),
              "tiny": (r.normal(size=(8, 48)) * 1e-6).astype(np.float32), "zeros": np.zeros((3, 17), np.float32)}
    settings = {
        "minifloat_ieee": [dict(width=8, exponent_width=4, exponent_bias=None), dict(width=8, exponent_width=4, exponent_bias=15),
                           dict(width=6, exponent_width=3, exponent_bias=2), dict(width=12, exponent_width=5, exponent_bias=None),
                           dict(width=5, exponent_width=4, exponent_bias=-3)],
        "minifloat_denorm": [dict(width=8, exponent_width=4, exponent_bias=None), dict(width=8, exponent_width=4, exponent_bias=15),
                             dict(width=6, exponent_width=3, exponent_bias=2), dict(width=12, exponent_width=5, exponent_bias=None),
                             dict(width=5, exponent_width=4, exponent_bias=-3)],
        "log": [dict(width=8, exponent_bias=None), dict(width=8, exponent_bias=100), dict(width=4, exponent_bias=3),
                dict(width=6, exponent_bias=-2)],
    }
    arrays, cases = {}, {}
    for iname, x in inputs.items():
        arrays[f"x/{iname}"] = x
    for qname, plist in settings.items():
        fn = refq.quantizers.QUANTIZER_MAP[qname]
        for pi, params in enumerate(plist):
            for iname, x in inputs.items():
                tag = f"{qname}/{pi}/{iname}"
                arrays[f"y/{tag}"] = fn(torch.from_numpy(x.copy()), **params).numpy()
                cases[tag] = {"quantizer": qname, "params": params, "input": iname}
    np.savez_compressed(OUT / "elementwise.npz", **arrays)
    (OUT / "elementwise.json").write_text(json.dumps(cases, indent=1))
    print(f"elementwise: {len(cases)} cases")


def gen_kat():
    """The two scalar known-answer values the reference's docstrings hold (minifloat.py:41-43,150-153)."""
    (OUT / "kat.json").write_text(json.dumps({
        "minifloat_ieee_8_4_bias15": {"bits": "1 0111 011", "value": -0.00537109375},
        "minifloat_denorm_8_4_bias15": {"bits": "1 0111 011", "value": -0.00146484375}}, indent=1))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(4)
    refq = _load_ref()
    gen_quantizers(refq)
    gen_modules(refq)
    gen_config_and_profile(refq)
    gen_elementwise(refq)
    gen_kat()


if __name__ == "__main__":
    main()
