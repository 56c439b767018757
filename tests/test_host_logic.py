"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares, the registry mirror has the reference's keys, the config parser and layer profiler
reproduce the reference's outputs (golden, captured by tools/gen_golden.py), and the block-shape
resolution agrees with the oracle's."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_c_abi_exports_every_declared_symbol():
    from mi355q import _lib
    header = (ROOT / "include" / "mi355q.h").read_text()
    declared = set(re.findall(r"\b(mi355q_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(str(_lib.library_path()))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mi355q.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert _lib.load_library().mi355q_abi_version() == int(re.search(r"MI355Q_ABI_VERSION (\d+)", header).group(1))
    assert int(re.search(r"MI355Q_WORKSPACE_BYTES (\d+)", header).group(1)) == _lib.WORKSPACE_BYTES


def test_bad_arguments_are_rejected_without_a_gpu():
    from mi355q import _lib
    lib = _lib.load_library()
    assert lib.mi355q_block_fp_quantize(None, None, None, None, 1, 4, 16, 1, 16, 6, 8, 127, 0, None, None) == -1
    assert lib.mi355q_block_fp_quantize(None, None, None, None, 1, 0, 16, 1, 16, 6, 8, 127, 0, None, None) == 0  # empty
    assert lib.mi355q_bfp_gemm(None, None, None, None, None, None, 4, 4, 24, 4, 5, 127, 5, 127, None) == -1
    assert b"bad argument" in lib.mi355q_error_string(-1)


def test_aligned_operand_entry_points_validate_without_a_gpu():
    """sizes of the aligned formats and the argument checks of the align / fused / GEMM entry points (nothing is
    launched: every call below is rejected, or empty, before it reaches the device)"""
    import ctypes as C
    from mi355q import _lib
    lib = _lib.load_library()
    E_BADARG, E_UNSUPPORTED, E_ALIGN = -1, -2, -3
    assert lib.mi355q_bfp_rows_pad(4096) == 4096 + 256 and lib.mi355q_bfp_rows_pad(1) == 512 and lib.mi355q_bfp_rows_pad(0) == 0
    assert lib.mi355q_bfp_tiled_bytes(100, 192) == 128 * 192 and lib.mi355q_bfp_tiled_bytes(0, 64) == 0
    assert lib.mi355q_bfp_row_list_bytes(4096, 0) == (8 + 16 * (8 + 8 * 120)) * 4 and lib.mi355q_bfp_row_list_bytes(1, 120) == (8 + 968) * 4
    assert lib.mi355q_bfp_row_list_bytes(512, 1016) == (8 + 2 * (8 + 8 * 1016)) * 4 and lib.mi355q_bfp_row_list_bytes(512, 1017) == 0
    one = C.create_string_buffer(64)
    p = C.addressof(one)
    # row alignment: K % 64, K <= 16384, pointers
    assert lib.mi355q_bfp_align_rows(p, p, p, p, p, p, None, 132, 4, 48, 0, None) == E_UNSUPPORTED
    assert lib.mi355q_bfp_align_rows(p, p, p, p, p, p, None, 132, 4, 16384 + 64, 0, None) == E_UNSUPPORTED
    assert lib.mi355q_bfp_align_rows(None, p, p, p, p, p, None, 132, 4, 64, 0, None) == E_BADARG
    assert lib.mi355q_bfp_align_rows(p, p, p, p, p, p, None, 132, 4, 64, 1017, None) == E_BADARG          # bucket_cap
    assert lib.mi355q_bfp_align_rows(p, p, p, p, p, p, None, 132, 0, 64, 0, None) == 0
    assert lib.mi355q_block_fp_quantize_aligned_rows(p, p, p, p, p, p, p, 4, 64, 6, 8, 127, 0, None) == E_BADARG   # list == list_to_clear
    assert lib.mi355q_block_fp_quantize_aligned_rows(p, p, p, p, p, p, None, 4, 64, 9, 8, 127, 0, None) == E_BADARG  # width
    assert lib.mi355q_block_fp_quantize_aligned_rows(p, p, p, p, p, p, None, 4, 64, 6, 8, 127, -2, None) == E_BADARG  # bucket_cap
    assert lib.mi355q_block_fp_quantize_aligned_rows(p, p, p, p, p, None, None, 4, 64, 6, 8, 127, 0, None) == E_BADARG   # list required unless NO_ALIGN
    assert lib.mi355q_block_fp_quantize_aligned_rows(p + 4, p, p, p, p, p, None, 4, 64, 6, 8, 127, 0, None) == E_ALIGN
    # GEMM: both operands in the same flavour, K % 64
    x, w = _lib.BfpOperand(p, p, p, p, p, 0, 5, 127, 1), _lib.BfpOperand(p, p, p, p, p, 8, 5, 127, 0)
    assert lib.mi355q_bfp_gemm_aligned(C.addressof(x), C.addressof(w), None, p, 4, 4, 128, 4, None) == E_BADARG
    w.row_aligned, w.list_cap = 1, 2000
    assert lib.mi355q_bfp_gemm_aligned(C.addressof(x), C.addressof(w), None, p, 4, 4, 128, 4, None) == E_BADARG       # bucket cap
    w.list_cap = 0
    x.row_aligned = w.row_aligned = 0
    assert lib.mi355q_bfp_gemm_aligned(C.addressof(x), C.addressof(w), None, p, 4, 4, 128, 4, None) == E_UNSUPPORTED     # group flavour: removed
    x.row_aligned = w.row_aligned = 1
    assert lib.mi355q_bfp_gemm_aligned(C.addressof(x), C.addressof(w), None, p, 4, 4, 48, 4, None) == E_UNSUPPORTED
    assert lib.mi355q_bfp_gemm_aligned(C.addressof(x), C.addressof(w), None, p, 0, 4, 128, 4, None) == 0
    assert lib.mi355q_bfp_gemm_aligned(C.addressof(x), C.addressof(w), None, p, 4, 4, 128, 2, None) == E_BADARG       # ldy < N
    assert lib.mi355q_block_fp_quantize_bf16(p, p, 1, 4, 64, 1, 16, 12, 8, 127, p, None) == E_UNSUPPORTED   # width > 9
    assert lib.mi355q_block_fp_quantize_bf16(p, None, 1, 4, 64, 1, 16, 6, 8, 127, p, None) == E_BADARG
    assert lib.mi355q_block_fp_quantize_bf16(p, p, 1, 0, 64, 1, 16, 6, 8, 127, p, None) == 0
    # fused quantise + matmul: blocks of 16 must tile K and N, widths must fit bf16, pointers
    assert lib.mi355q_bfp_matmul_workspace_bytes(12, 2048, 64) == 12 * 2048 * 64 * 2 + 64 and lib.mi355q_bfp_matmul_workspace_bytes(1, 80, 16) == 128 * 16 * 2 + 64 and lib.mi355q_bfp_matmul_workspace_bytes(0, 4, 4) == 0
    mm = lambda *a: lib.mi355q_bfp_matmul(*a)
    assert mm(p, p, p, p, 2, 8, 24, 16, 6, 8, 127, 6, 8, 127, None) == E_UNSUPPORTED      # K % 16
    assert mm(p, p, p, p, 2, 8, 32, 10, 6, 8, 127, 6, 8, 127, None) == E_UNSUPPORTED      # N % 16
    assert mm(p, p, p, p, 2, 8, 32, 16, 12, 8, 127, 6, 8, 127, None) == E_UNSUPPORTED     # width > 9
    assert mm(p, p, p, None, 2, 8, 32, 16, 6, 8, 127, 6, 8, 127, None) == E_BADARG        # workspace
    assert mm(p, p, p, p, 2, 8, 32, 16, 6, 9, 127, 6, 8, 127, None) == E_BADARG           # exponent width
    assert mm(p, p + 4, p, p, 2, 8, 32, 16, 6, 8, 127, 6, 8, 127, None) == E_ALIGN
    assert mm(p, p, p, p, 0, 8, 32, 16, 6, 8, 127, 6, 8, 127, None) == 0


def test_registry_keys_match_reference():
    import mi355q.quantize as Q
    names = {"block_fp", "block_log", "block_minifloat", "integer", "log", "minifloat_denorm", "minifloat_ieee"}
    assert set(Q.QUANTIZER_MAP) == names
    assert set(Q.QUANTIZED_MODULE_MAP) == {"linear"} and set(Q.QUANTIZED_MODULE_MAP["linear"]) == names
    # the reference's three ops, plus this build's two additions (the attention's softmax stage folded into the product)
    ref_ops = {"matmul", "bmm", "rotary_positional_encoding"}
    assert set(Q.QUANTIZED_FUNC_MAP) == ref_ops | {"softmax_matmul", "softmax_bmm", "attention"}
    for name in ref_ops:
        assert set(Q.QUANTIZED_FUNC_MAP[name]) == names
    assert set(Q.QUANTIZED_FUNC_MAP["softmax_bmm"]) == set(Q.QUANTIZED_FUNC_MAP["softmax_matmul"]) == {"block_fp", "block_minifloat"}
    # reference quirk: "log" is served by the block_log functions
    assert Q.QUANTIZED_FUNC_MAP["matmul"]["log"] is Q.QUANTIZED_FUNC_MAP["matmul"]["block_log"]
    assert Q.get_quantized_cls("linear", {"name": "block_fp"}).__name__ == "LinearBlockFP"
    assert Q.get_quantizer("x", {"name": "block_log"}) is Q.QUANTIZER_MAP["block_log"]


def test_parse_node_config_golden(golden_config_profile):
    from mi355q.quantize import parse_node_config
    cases = golden_config_profile["parse_node_config"]
    assert len(cases) > 50
    for key, c in cases.items():
        op = key.split(":")[-1]
        if "raises" in c:
            with pytest.raises(Exception) as ei:
                parse_node_config(dict(c["in"]), op, strict=True)
            assert type(ei.value).__name__ == c["raises"], key
        else:
            assert parse_node_config(dict(c["in"]), op, strict=True) == c["out"], key


def test_layer_profiler_golden(golden_config_profile):
    from mi355q.quantize import profile_linear_layer, profile_matmul_layer, update_profile
    for fn, key in ((profile_linear_layer, "profile_linear_layer"), (profile_matmul_layer, "profile_matmul_layer")):
        for c in golden_config_profile[key]:
            args = [tuple(a) if isinstance(a, list) else a for a in c["args"]]
            if "raises" in c:
                with pytest.raises(Exception) as ei:
                    fn(dict(c["cfg"]), *args)
                assert type(ei.value).__name__ == c["raises"]
            else:
                got = fn(dict(c["cfg"]), *args)
                assert {k: int(v) for k, v in got.items()} == c["out"], (key, c["args"])
    p = {k: 0 for k in ("num_params", "num_acts", "param_bits", "act_bits", "flops")}
    d = {k: 3 for k in p}
    assert update_profile(update_profile(p, d), d) == {k: 6 for k in p}


@pytest.mark.parametrize("shape,block,skip", [
    ((50,), [16], False), ((12, 80), [1, 16], True), ((5, 43), [1, 16], True), ((6, 40), [16], True),
    ((24, 64), [1, 16], False), ((10, 24), [4, 8], False), ((6, 40), [16], False), ((3, 7, 48), [1, 16], True),
    ((2, 6, 40), [2, 16], True), ((2, 5, 40), [16], True), ((4, 10), [1, 16], True), ((4, 10), 16, True),
    ((7, 9), [3, 1, 16], False)])
def test_resolve_blocking_matches_oracle(shape, block, skip):
    from mi355q.ops import n_blocks, resolve_blocking
    from oracle.np_oracle import block_meta
    lead, rows, cols, b0, b1 = resolve_blocking(shape, block, skip)
    m = block_meta(shape, block, skip)
    assert (b0, b1) == (m.b0, m.b1)
    assert n_blocks(lead, rows, cols, b0, b1) == m.n_blocks
    assert lead * rows * cols == int(np.prod(shape))


def test_resolve_blocking_errors_like_reference():
    from mi355q.ops import resolve_blocking
    with pytest.raises(NotImplementedError):
        resolve_blocking((2, 3, 4), [1, 16], False)
    with pytest.raises(RuntimeError):
        resolve_blocking((2, 3, 4, 5), [1, 16], True)
    with pytest.raises(AssertionError):
        resolve_blocking((8,), [16], True)


def test_linear_contract_on_cpu():
    """construction, attributes, repr, from_float, bypass forward work without a GPU; the quantised
    forward refuses a CPU tensor instead of silently computing elsewhere."""
    import torch
    import mi355q.quantize as Q
    cfg = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8,
               data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=4, weight_exponent_width=8,
               weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8,
               bias_exponent_bias=127, bias_block_size=[16])
    cls = Q.get_quantized_cls("linear", cfg)
    fp = torch.nn.Linear(32, 16)
    lin = cls.from_float(fp, cfg)
    assert isinstance(lin, torch.nn.Linear) and isinstance(lin.weight, torch.nn.Parameter)
    assert torch.equal(lin.weight, fp.weight) and torch.equal(lin.bias, fp.bias)
    assert lin.is_ptq and lin.weight_requires_quantisation and not lin.bypass
    assert lin.x_quantizer.keywords["skip_first_dim"] is True and lin.w_quantizer.keywords["skip_first_dim"] is False
    assert lin.x_quantizer.keywords["width"] == 6 and lin.w_quantizer.keywords["width"] == 4
    assert repr(lin) == ("LinearBlockFP(in_features=32, out_features=16, bias=True, bypass=False, is_ptq=True, "
                         "x/w/b-width=6/4/6)")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lin(torch.zeros(2, 32))
    byp = cls(32, 16, config=dict(cfg, bypass=True))
    x = torch.randn(3, 32)
    assert torch.equal(byp(x), torch.nn.functional.linear(x, byp.weight, byp.bias))
    with pytest.raises(KeyError):
        cls(32, 16, config={"name": "block_fp", "is_ptq": True})


def test_no_kernel_uses_scratch_memory(tmp_path):
    """Every gfx950 kernel in libmi355q.so keeps its state in registers / LDS: private_segment_fixed_size == 0.

    A kernel that touches scratch memory pays for it on every launch (a by-value copy of the argument block indexed at
    run time once put the tile GEMM there: +13 us per launch, 94 -> 114 us on the headline step), and the compiler
    does that silently, so the built code objects are checked here.
    """
    import shutil
    import subprocess

    from mi355q import _lib

    llvm = Path("/opt/rocm/lib/llvm/bin")
    tools = [llvm / "llvm-objcopy", llvm / "clang-offload-bundler", llvm / "llvm-readelf"]
    if not all(t.exists() for t in tools) or not Path(_lib.library_path()).exists():
        pytest.skip("ROCm llvm tools or the built library are not here")
    fat = tmp_path / "fat.bin"
    subprocess.run([str(tools[0]), "-O", "binary", "--only-section=.hip_fatbin", str(_lib.library_path()), str(fat)], check=True)
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundles in .hip_fatbin"
    kernels, offenders = 0, []
    for i, s in enumerate(starts):
        part = tmp_path / f"bundle{i}.bin"
        part.write_bytes(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"co{i}.o"
        subprocess.run([str(tools[1]), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={part}", f"--output={co}"], check=True)
        notes = subprocess.run([str(tools[2]), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            line = line.strip()
            if line.startswith(".name:"):
                name = line.split(":", 1)[1].strip()
            elif line.startswith(".private_segment_fixed_size:"):
                kernels += 1
                size = int(line.split(":", 1)[1])
                if size:
                    offenders.append((name, size))
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert kernels >= 100, kernels
    assert not offenders, offenders


def test_pre_op_entry_points_validate_without_a_gpu():
    """the quantisers with a folded elementwise step reject unknown ops and a missing / misaligned second input"""
    from mi355q import _lib
    lib = _lib.load_library()
    E_BADARG = _lib.E_BADARG
    p = 1 << 20
    assert lib.mi355q_block_fp_quantize_aligned_rows_pre(p, None, 3, p, p, p, p, p, None, 4, 64, 6, 8, 127, 0, None) == E_BADARG
    assert lib.mi355q_block_fp_quantize_aligned_rows_pre(p, None, 2, p, p, p, p, p, None, 4, 64, 6, 8, 127, 0, None) == E_BADARG
    assert lib.mi355q_block_fp_quantize_aligned_rows_pre(p, p + 4, 2, p, p, p, p, p, None, 4, 64, 6, 8, 127, 0, None) == E_BADARG
    assert lib.mi355q_block_fp_quantize_bf16_tiled_pre(p, None, 2, None, p, 4, 64, 6, 8, 127, p, None) == E_BADARG
    assert lib.mi355q_block_fp_quantize_bf16_tiled_pre(p, None, -1, None, p, 4, 64, 6, 8, 127, p, None) == E_BADARG
    assert lib.mi355q_block_fp_quantize_bf16_tiled_pre(p, None, 1, None, p, 0, 64, 6, 8, 127, p, None) == 0       # nothing to do


def test_round6_entry_points_validate_without_a_gpu():
    """the entry points of ABI 24 reject what they do not take before anything is launched (include/mi355q.h): the split-bf16 operand
    builder, the fused attention pass, the int8 product with a residual"""
    import ctypes
    from mi355q import _lib
    lib = _lib.load_library()
    BAD, UNS, ALIGN = _lib.E_BADARG, _lib.E_UNSUPPORTED, _lib.E_ALIGN
    p = 1 << 20
    assert lib.mi355q_abi_version() == _lib.ABI_VERSION >= 24
    # mi355q_fp32_split_tile(x, y, rows, K, role, stream)
    assert lib.mi355q_fp32_split_tile(p, p, 4, 64, 2, None) == BAD                  # roles are 0 (left) and 1 (right)
    assert lib.mi355q_fp32_split_tile(None, p, 4, 64, 0, None) == BAD
    assert lib.mi355q_fp32_split_tile(p, p, 4, 48, 0, None) == UNS                  # K % 32
    assert lib.mi355q_fp32_split_tile(p + 4, p, 4, 64, 0, None) == ALIGN
    assert lib.mi355q_fp32_split_tile(p, p, 0, 64, 0, None) == 0                    # nothing to do
    # mi355q_bfp_gemm_aligned_res(x, w, bias, residual, ldr, y, M, N, K, ldy, stream)
    assert lib.mi355q_bfp_gemm_aligned_res(p, p, None, None, 64, p, 4, 64, 128, 64, None) == BAD      # no residual
    assert lib.mi355q_bfp_gemm_aligned_res(p, p, None, p, 32, p, 4, 64, 128, 64, None) == BAD        # ldr < N
    assert lib.mi355q_bfp_gemm_aligned_res(p, p, None, p, 66, p, 4, 64, 128, 66, None) == BAD        # ldr % 4
    assert lib.mi355q_bfp_gemm_aligned_res(p, p, None, p + 4, 64, p, 4, 64, 128, 64, None) == ALIGN
    # mi355q_bfp_attention_fused(q, k, v, mask, causal, q_scale, scale_div, out, out_tiled, consumer, ws, B, M, T, D, qk, pv, strides,
    #                            cos, sin, pos, table_rows, heads, stream)
    par = (ctypes.c_int32 * 6)(6, 8, 127, 6, 8, 127)
    pa = ctypes.addressof(par)
    call = lambda **kw: lib.mi355q_bfp_attention_fused(*[kw.get(n, d) for n, d in (
        ("q", p), ("k", p), ("v", p), ("mask", None), ("causal", 1), ("q_scale", 0.0), ("scale_div", 0.0), ("out", p), ("tiled", None), ("cons", None),
        ("ws", p), ("B", 4), ("M", 64), ("T", 64), ("D", 128), ("qk", pa), ("pv", pa), ("strides", None), ("cos", None), ("sin", None), ("pos", None),
        ("rows", 0), ("heads", 1), ("stream", None))])
    assert call(cos=p) == BAD                                                       # tables and positions come together
    assert call(cos=p, sin=p, pos=p, rows=0, heads=4) == BAD                        # an empty table
    assert call(cos=p, sin=p + 4, pos=p, rows=64, heads=4) == ALIGN
    assert call(cos=p, sin=p, pos=p, rows=64, heads=4, D=96) == UNS                 # the embedding on load: head_dim 64 / 128
    assert call(cos=p, sin=p, pos=p, rows=64, heads=4, M=32) == UNS                 # ... over the same positions for q and k
    assert call(tiled=p) == BAD                                                     # the consumer's quantiser is missing
    cons = (ctypes.c_int32 * 3)(10, 8, 127)
    assert call(tiled=p, cons=ctypes.addressof(cons)) == UNS                        # wider than bf16 holds exactly
    cons = (ctypes.c_int32 * 3)(6, 8, 127)
    assert call(tiled=p, cons=ctypes.addressof(cons), mask=p) == UNS                # the operand output: no additive mask
    assert call(tiled=p, cons=ctypes.addressof(cons), D=32) == UNS
    assert call(q_scale=0.125, D=32) == UNS                                         # q_scale rides the fragment pack: head_dim 64 / 128
    assert call(q_scale=0.125, M=128, causal=0) == UNS                              # ... and no more queries than keys
    assert call(B=0) == 0                                                           # nothing to do


def test_stream_cache_evicts_per_stream_and_never_a_captured_streams_buffers():
    """ops._StreamCache: least-recently-used per (device, stream); entries of a stream that recorded a HIP graph
    are never dropped, whatever traffic other streams (or that stream) see afterwards"""
    from mi355q.ops import _StreamCache
    c = _StreamCache(4)
    eager, graph = (0, 111), (0, 222)
    for i in range(3):
        c.put(graph + (i,), f"g{i}")
    _StreamCache.pin_stream(*graph)
    try:
        for i in range(40):                       # variable-length prompts on the eager stream
            c.put(eager + (i,), i)
            if i >= 1:
                assert c.get(eager + (i - 1,)) == i - 1       # the one just used is still there
        assert sum(1 for k in c.keys() if k[:2] == eager) == 4
        assert c.get(eager + (0,)) is None and c.get(eager + (39,)) == 39
        assert [c.get(graph + (i,)) for i in range(3)] == ["g0", "g1", "g2"]
        for i in range(3, 20):                    # more shapes on the pinned stream: kept, all of them
            c.put(graph + (i,), f"g{i}")
        assert sum(1 for k in c.keys() if k[:2] == graph) == 20
        # use refreshes: the oldest-used entry goes first
        d = _StreamCache(2)
        d.put((0, 1, "a"), 1), d.put((0, 1, "b"), 2)
        d.get((0, 1, "a"))
        d.put((0, 1, "c"), 3)
        assert d.get((0, 1, "b")) is None and d.get((0, 1, "a")) == 1
    finally:
        _StreamCache._pinned_streams.discard(graph)


def test_mask_2d_accepts_only_masks_shared_by_every_leading_index():
    import torch
    from mi355q.quantize.quantized_functions import _mask_2d
    T = 8
    assert _mask_2d(torch.zeros(1, 1, T, T), T, T).shape == (T, T)
    assert _mask_2d(torch.zeros(T, T), T, T).shape == (T, T)
    m = _mask_2d(torch.arange(T, dtype=torch.float32).reshape(1, 1, 1, T), 4, T)       # padding mask: one row for all
    assert m.shape == (4, T) and m.is_contiguous() and torch.equal(m[3], torch.arange(T, dtype=torch.float32))
    assert _mask_2d(torch.zeros(2, 1, T, T), T, T) is None                              # per-batch
    assert _mask_2d(torch.zeros(1, 1, T, 4), T, T) is None                              # does not broadcast
    assert _mask_2d(torch.zeros(T), T, T) is None
