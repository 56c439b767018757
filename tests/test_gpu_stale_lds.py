"""No kernel of the path reads LDS it has not written: LDS keeps what the previous kernel on the compute unit left
there, so a read of an unwritten word shows up as a result that depends on that.  A helper kernel
(tools/lds_poison) fills all 160 KiB of every compute unit with a pattern in front of each call; every output must be
the same bits under every pattern (zeros, all ones, NaN, 1.0f, denormal, sign bit)."""
import ctypes
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
PATTERNS = (0x00000000, 0xFFFFFFFF, 0x7FC00000, 0x3F800000, 0x00000001, 0x80000000)


@pytest.fixture(scope="module")
def poison():
    import torch
    so = ROOT / "tools" / "lds_poison" / "liblds_poison.so"
    if not so.exists():
        pytest.fail("tools/lds_poison/liblds_poison.so is not built (__graft_entry__.build())")
    lib = ctypes.CDLL(str(so))

    def fill(pattern):
        rc = lib.lds_poison(ctypes.c_uint(pattern), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    return fill


def test_the_screen_sees_stale_lds(poison):
    """the screen itself: a kernel that reads LDS it never wrote sees the pattern left by the fill (else the tests below
    would pass vacuously)"""
    import torch
    lib = ctypes.CDLL(str(ROOT / "tools" / "lds_poison" / "liblds_poison.so"))
    out = torch.zeros(1024, dtype=torch.int32, device="cuda:0")
    for p in (0x3F800000, 0x12345678):
        poison(p)
        rc = lib.lds_peek(ctypes.c_void_p(out.data_ptr()), ctypes.c_int(1024), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        seen = out.cpu()
        assert (seen == p).float().mean().item() > 0.9, f"stale LDS not visible: {seen[:8].tolist()}"


def _same_under_every_pattern(poison, fn):
    import torch
    outs = []
    for p in PATTERNS:
        poison(p)
        out = fn()
        outs.append([o.clone() for o in (out if isinstance(out, (tuple, list)) else (out,))])
    torch.cuda.synchronize()
    for p, o in zip(PATTERNS[1:], outs[1:]):
        for a, b in zip(outs[0], o):
            assert torch.equal(a.view(torch.uint8) if a.dtype != torch.bool else a, b.view(torch.uint8) if b.dtype != torch.bool else b), \
                f"output depends on stale LDS (pattern {p:#010x})"


def _lin_cfg(width, **extra):
    return dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=width, data_in_exponent_width=8,
                data_in_exponent_bias=127, data_in_block_size=[1, 16], weight_width=width, weight_exponent_width=8,
                weight_exponent_bias=127, weight_block_size=[1, 16], bias_width=width, bias_exponent_width=8,
                bias_exponent_bias=127, bias_block_size=[16], **extra)


@pytest.mark.parametrize("M,K,N,align,act", [
    (1024, 1024, 1024, "rows", "plain"),        # 128-row tiles, in-tile exception vectors
    (300, 512, 256, "rows", "plain"),           # 128-row tiles, few K-steps: the buckets must have landed before they are read
    (256, 256, 256, "rows", "plain"),           #   (round 4: the pipelined schedule of that tile waited for too little)
    (4096, 2048, 2048, "rows", "plain"),        # 256-row tiles
    (256, 4096, 256, "rows", "uniform"),        # split-K slices
    (512, 2048, 512, "rows", "overflow"),       # bucket overflow -> the blockwise product inside the launch
    (1024, 1024, 1024, "auto", "silu"),         # per-block-exponent route (bf16 tile GEMM)
    (300, 512, 272, "blocks", "plain"),         # every block keeps its exponent (bf16 tile GEMM)
])
def test_linear_routes(poison, M, K, N, align, act):
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    dev = "cuda:0"
    torch.manual_seed(M + K)
    cfg = _lin_cfg(6, mi355q_align=align)
    lin = Q.get_quantized_cls("linear", cfg)(K, N, bias=True, config=cfg).to(dev)
    x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
    if act == "uniform":
        x = (torch.rand(M, K, device=dev) * 0.9 + 0.6) * torch.sign(x)
    if act == "silu":
        x = torch.nn.functional.silu(x) * torch.randn(M, K, device=dev)
    with torch.no_grad():
        lin.weight[::7, 48:64] *= 2.0 ** 6
        x[::9, 32:48] *= 2.0 ** -9
        if act == "overflow":
            x[:64].view(64, K // 16, 16)[:, ::2] *= 2.0 ** -8
        lin(x)                                               # first PTQ forward packs the weight
        old, ops.REUSE_QUANTISED_INPUT = ops.REUSE_QUANTISED_INPUT, False
        try:
            if act == "overflow":
                # (the blockwise product's atomic add-back is order-dependent in the last bit: compare loosely)
                outs = []
                for p in PATTERNS:
                    poison(p)
                    outs.append(lin(x).clone())
                for o in outs[1:]:
                    torch.testing.assert_close(o, outs[0], rtol=1e-5, atol=1e-5)
            else:
                _same_under_every_pattern(poison, lambda: lin(x))
        finally:
            ops.REUSE_QUANTISED_INPUT = old


@pytest.mark.parametrize("name,kw", [
    ("block_fp", dict(width=6, exponent_width=8, exponent_bias=None, block_size=[1, 16])),
    ("block_minifloat", dict(width=6, exponent_width=2, exponent_bias_width=8, block_size=[1, 16])),
    ("block_log", dict(width=6, exponent_bias_width=8, block_size=[1, 16])),
    ("minifloat_ieee", dict(width=8, exponent_width=4, exponent_bias=None)),
    ("log", dict(width=8, exponent_bias=None)),
])
def test_quantisers(poison, name, kw):
    import torch
    import mi355q.quantize as Q
    torch.manual_seed(3)
    x = torch.randn(777, 1040, device="cuda:0") * torch.exp(2 * torch.randn(777, 1, device="cuda:0"))
    x[5] = 0
    q = Q.get_quantizer("", dict(name=name))
    _same_under_every_pattern(poison, lambda: q(x, **kw))


@pytest.mark.parametrize("heads,T,D,kernel", [(4, 512, 64, 0), (2, 1536, 64, 3), (2, 640, 128, 1), (2, 2304, 64, 2)])
def test_attention(poison, heads, T, D, kernel):
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    torch.manual_seed(heads + T)
    dev = "cuda:0"
    q, k, v = (torch.randn(heads, T, D, device=dev) for _ in range(3))
    cfg = dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=None,
               data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=None,
               weight_block_size=[1, 16])
    f = Q.get_quantized_func("attention", cfg)
    prev = ops.attention_set_kernel(kernel)
    try:
        _same_under_every_pattern(poison, lambda: f(q, k, v, cfg, cfg, causal=True, scale_div=float(D) ** 0.5))
    finally:
        ops.attention_set_kernel(prev)


def test_attention_products_and_rope(poison):
    import torch
    import mi355q.quantize as Q
    from mi355q import ops
    torch.manual_seed(9)
    dev = "cuda:0"
    cfg = dict(name="block_fp", bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=None,
               data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=None,
               weight_block_size=[1, 16])
    a = torch.randn(6, 384, 64, device=dev)
    b = torch.randn(6, 64, 384, device=dev)
    bmm = Q.get_quantized_func("bmm", cfg)
    _same_under_every_pattern(poison, lambda: bmm(a, b, cfg))
    s = torch.randn(6, 384, 384, device=dev)
    v = torch.randn(6, 384, 64, device=dev)
    sm = Q.get_quantized_func("softmax_bmm", cfg)
    _same_under_every_pattern(poison, lambda: sm(s, v, cfg, causal=True))
