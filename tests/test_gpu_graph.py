"""HIP-graph replay of the quantised forward == eager forward, bit for bit (mi355q/graphs.py)."""
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "llm-mixed-q_amd"))

pytestmark = pytest.mark.gpu

W6A6 = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
            data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
            weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])


def _model(family, fused):
    from mi355q.harness import (TinyLlamaConfig, TinyLlamaForCausalLM, TinyOPTConfig, TinyOPTForCausalLM,
                                expand_llama_quant_config, expand_quant_config)
    torch.manual_seed(3)
    qc = dict(W6A6, mi355q_fused_softmax=fused)
    if family == "opt":
        cfg = TinyOPTConfig(vocab_size=512, hidden_size=256, ffn_dim=1024, num_layers=2, num_heads=4, max_positions=512)
        m = TinyOPTForCausalLM(cfg, expand_quant_config(qc, cfg.num_layers))
    else:
        cfg = TinyLlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, max_positions=512)
        m = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(qc, cfg.num_layers))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 2 and "embed" not in n:
                p.mul_(3.0)
    return m.to("cuda:0").eval(), cfg


@pytest.mark.parametrize("family,fused", [("opt", True), ("opt", False), ("llama", True)])
def test_graph_replay_equals_eager(family, fused):
    from mi355q.graphs import GraphedForward
    model, cfg = _model(family, fused)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    ids = [torch.randint(0, cfg.vocab_size, (1, 384), generator=g).to(dev) for _ in range(3)]
    with torch.no_grad():
        eager = [model(i)[0].clone() for i in ids]
    fwd = GraphedForward(lambda t: model(t)[0], (ids[0],))
    for rep in range(2):                                  # the second round replays nodes whose lists were used before
        for i, e in zip(ids, eager):
            out = fwd(i)
            assert torch.equal(out, e), (family, fused, rep, float((out - e).abs().max()))
    with torch.no_grad():                                 # eager calls after replays still see clean state
        for i, e in zip(ids, eager):
            assert torch.equal(model(i)[0], e)


def test_graph_linear_with_exceptions():
    """a row-scale Linear whose activations carry exception blocks: the captured node's own list is cleared on every replay"""
    from mi355q.graphs import GraphedForward
    from mi355q.quantize import get_quantized_cls
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    lin = get_quantized_cls("linear", W6A6)(512, 256, bias=True, config=dict(W6A6, mi355q_align="rows")).to(dev)
    xs = []
    for s in range(3):
        x = torch.randn(300, 512, generator=torch.Generator().manual_seed(s))
        x[::7, 32:48] *= 2.0 ** -9                        # blocks far below their row's window: exceptions
        x[5::11, 64:80] *= 2.0 ** 7
        xs.append(x.to(dev))
    with torch.no_grad():
        eager = [lin(x).clone() for x in xs]
    fwd = GraphedForward(lambda t: lin(t), (xs[0],))
    for rep in range(3):
        for x, e in zip(xs, eager):
            assert torch.equal(fwd(x), e), rep
    with torch.no_grad():
        for x, e in zip(xs, eager):
            assert torch.equal(lin(x), e)


@pytest.mark.parametrize("family", ["opt", "llama"])
def test_graph_replay_with_every_knob(family):
    """the forward with every implementation knob on (norms and activations inside the quantisers, grouped projections,
    one-pass attention writing token-major) captured into a HIP graph: replays == the eager forward, bit for bit, also on
    new token ids"""
    import torch
    from mi355q import harness as H
    from mi355q.graphs import GraphedForward
    qc = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
              data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
              weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
              mi355q_fused_attention=True, mi355q_token_major_output=True, mi355q_grouped_linear=True,
              mi355q_fused_activation=True, mi355q_fused_norm=True)
    torch.manual_seed(3)
    if family == "opt":
        cfg = H.TinyOPTConfig(vocab_size=512, hidden_size=256, ffn_dim=1024, num_layers=3, num_heads=4, max_positions=512)
        m = H.TinyOPTForCausalLM(cfg, H.expand_quant_config(qc, cfg.num_layers))
    else:
        cfg = H.TinyLlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=768, num_layers=3, num_heads=4, max_positions=512)
        m = H.TinyLlamaForCausalLM(cfg, H.expand_llama_quant_config(qc, cfg.num_layers))
    m = m.to("cuda:0").eval()
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, cfg.vocab_size, (1, 384), generator=g).to("cuda:0")
    ids2 = torch.randint(0, cfg.vocab_size, (1, 384), generator=g).to("cuda:0")
    with torch.no_grad():
        m(ids)
        want, want2 = m(ids)[0].clone(), m(ids2)[0].clone()
        fwd = GraphedForward(lambda t: m(t)[0], (ids,))
        assert torch.equal(fwd(ids), want)
        assert torch.equal(fwd(ids2), want2)
        assert torch.equal(fwd(ids), want)


def test_split_k_shape_first_seen_under_capture_and_eager_traffic_between_replays():
    """(1) a split-K GEMM shape whose workspace does not exist yet when a stream starts recording launches
    unsplit instead of allocating under capture; (2) eager calls with many other shapes between replays neither evict the
    buffers the graph replays into nor make the reuse record take a replayed output for the previous one"""
    from mi355q.graphs import GraphedForward
    from mi355q.quantize import get_quantized_cls
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    lin = get_quantized_cls("linear", W6A6)(4096, 512, bias=False, config=dict(W6A6, mi355q_align="rows")).to(dev)   # 2 x 2 tiles: split
    x = torch.randn(512, 4096, device=dev)
    with torch.no_grad():
        lin(torch.randn(64, 4096, device=dev))            # PTQ weight work + route decision outside the graph, another shape
    st = torch.cuda.Stream(device=dev)                     # a stream the library has no workspace for
    st.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st), torch.no_grad():
        y = lin(x)
    graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        assert torch.equal(y, lin(x))

    tail = get_quantized_cls("linear", W6A6)(512, 256, bias=False, config=dict(W6A6, mi355q_align="rows")).to(dev)
    xs = [torch.randn(512, 4096, generator=torch.Generator().manual_seed(s)).to(dev) for s in range(3)]
    with torch.no_grad():
        want = [tail(lin(t)).clone() for t in xs]
    fwd = GraphedForward(lambda t: lin(t), (xs[0],))
    for rep in range(2):
        for t, w in zip(xs, want):
            h = fwd(t)
            with torch.no_grad():
                assert torch.equal(tail(h), w), rep        # (an eager Linear on the graph's static output: never a stale hit)
                for rows in range(16, 16 * 40, 16):        # variable-length eager traffic on the default stream
                    lin(torch.randn(rows, 4096, device=dev))


@pytest.mark.parametrize("arith", ["block_minifloat", "block_log"])
def test_graph_replay_block_minifloat_and_block_log_models(arith):
    """a Llama-style model under block_minifloat / block_log (Linear layers on the bf16 tile GEMM, block_minifloat's attention
    products as bf16 operands on bf16 MFMAs, the streaming quantisers with their zero-block state -- block_log's fix-up
    launches and the fill word they leave from call to call, causal probabilities included): HIP-graph replay == eager, bit
    for bit"""
    from mi355q.graphs import GraphedForward
    from mi355q.harness import TinyLlamaConfig, TinyLlamaForCausalLM, expand_llama_quant_config
    if arith == "block_minifloat":
        d = dict(name="block_minifloat", is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_width=4,
                 data_in_exponent_bias_width=8, data_in_block_size=[1, 16], weight_width=8, weight_exponent_width=4,
                 weight_exponent_bias_width=8, weight_block_size=[1, 16], bias_width=8, bias_exponent_width=4,
                 bias_exponent_bias_width=8, bias_block_size=[16])
    else:
        d = dict(name="block_log", is_ptq=True, bypass=False, data_in_width=8, data_in_exponent_bias_width=8,
                 data_in_block_size=[1, 16], weight_width=8, weight_exponent_bias_width=8, weight_block_size=[1, 16],
                 bias_width=8, bias_exponent_bias_width=8, bias_block_size=[16])
    torch.manual_seed(2)
    cfg = TinyLlamaConfig(vocab_size=384, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, max_positions=128)
    model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(d, cfg.num_layers))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.ndim == 2 and "embed" not in n:
                p.mul_(40.0 if arith == "block_minifloat" else 4.0)   # (quirk 5: minifloat blocks below 2 quantise to zeros)
    model = model.to("cuda:0").eval()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    ids = [torch.randint(0, cfg.vocab_size, (1, 96), generator=g).to(dev) for _ in range(3)]
    with torch.no_grad():
        eager = [model(i)[0].clone() for i in ids]
    assert float(eager[0].abs().max()) > 0 and not torch.equal(eager[0], eager[1])
    fwd = GraphedForward(lambda t: model(t)[0], (ids[0],))
    for rep in range(2):
        for i, e in zip(ids, eager):
            assert torch.equal(fwd(i), e), rep
