"""Where do the Llama-style harness and the oracle part ways?  Layer-0 intermediates, GPU vs numpy oracle."""
import sys, math
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import torch.nn.functional as F
from mi355q.harness import TinyLlamaConfig, TinyLlamaForCausalLM, expand_llama_quant_config
from mi355q.quantize import get_quantized_func
from oracle import np_oracle as O
d = dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
         data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
         weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
torch.manual_seed(0)
cfg = TinyLlamaConfig(vocab_size=2048, hidden_size=768, intermediate_size=2048, num_layers=1, num_heads=12, max_positions=2048)
model = TinyLlamaForCausalLM(cfg, expand_llama_quant_config(d, 1))
with torch.no_grad():
    for n, p in model.named_parameters():
        if p.ndim == 2 and "embed" not in n: p.mul_(2.0)
sd = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()}
T = 512
ids = torch.randint(0, cfg.vocab_size, (1, T))
dev = torch.device("cuda:0")
model = model.to(dev)
L = model.layers[0]; A = L.self_attn
def rel(name, g, o):
    g = g.detach().cpu().numpy().astype(np.float64); o = np.asarray(o, np.float64)
    print(f"{name:12s} max|diff| / max|ref| = {np.abs(g - o).max() / (np.abs(o).max() + 1e-30):.3e}   differing elements {(g != o).mean():.4f}")
with torch.no_grad():
    x = model.embed_tokens(ids.to(dev))
    xo = sd["embed_tokens.weight"][ids.numpy()]
    h = L.input_layernorm(x)
    v_ = (xo.astype(np.float32) ** 2).mean(-1, keepdims=True)
    ho = (sd["layers.0.input_layernorm.weight"] * (xo * (np.float32(1) / np.sqrt(v_ + np.float32(cfg.rms_eps)))).astype(np.float32)).astype(np.float32)
    rel("rmsnorm", h, ho)
    B, nh, hd = 1, 12, 64
    q = A.q_proj(h); qo = O.linear_ptq(ho, sd["layers.0.self_attn.q_proj.weight"], None, d)[0]
    rel("q_proj", q, qo)
    qo_same_in = O.linear_ptq(h.cpu().numpy(), sd["layers.0.self_attn.q_proj.weight"], None, d)[0]
    rel("q_proj|same", q, qo_same_in)
    k = A.k_proj(h); vv = A.v_proj(h)
    sh = lambda t: t.view(B, T, nh, hd).transpose(1, 2)
    pos = torch.arange(T, device=dev)[None]
    rc = A.qc["rotary_positional_encoding"]
    qr, kr = get_quantized_func("rotary_positional_encoding", rc)(sh(q), sh(k), A.cos[:, :, :T], A.sin[:, :, :T], pos, config=rc)
    kw = {kk: d[f"data_in_{kk}"] for kk in ("width", "exponent_width", "exponent_bias", "block_size")}
    cos = O.block_fp_quantize(A.cos.cpu().numpy()[0, 0, :T], **kw, skip_first_dim=False)[None, None]
    sin = O.block_fp_quantize(A.sin.cpu().numpy()[0, 0, :T], **kw, skip_first_dim=False)[None, None]
    rh = lambda t: np.concatenate([-t[..., hd // 2:], t[..., :hd // 2]], -1)
    qn = sh(q).cpu().numpy(); kn = sh(k).cpu().numpy()
    rel("rope q|same", qr, (qn * cos + rh(qn) * sin).astype(np.float32))
    w = get_quantized_func("matmul", A.qc["matmul_0"])(qr, kr.transpose(2, 3), config=A.qc["matmul_0"])
    wo = O.matmul_quantized(qr.cpu().numpy(), np.ascontiguousarray(kr.cpu().numpy().transpose(0, 1, 3, 2)), d)
    rel("qk^T|same", w, wo)
    mask = torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)[None, None]
    w2 = torch.max(w / math.sqrt(hd) + mask, torch.tensor(torch.finfo(torch.float32).min, device=dev))
    p = F.softmax(w2, dim=-1, dtype=torch.float32)
    wn = w2.cpu().numpy(); wn = wn - wn.max(-1, keepdims=True)
    pn = (np.exp(wn) / np.exp(wn).sum(-1, keepdims=True)).astype(np.float32)
    rel("softmax|same", p, pn)
    o = get_quantized_func("matmul", A.qc["matmul_1"])(p, sh(vv), config=A.qc["matmul_1"])
    oo = O.matmul_quantized(p.cpu().numpy(), np.ascontiguousarray(sh(vv).cpu().numpy()), d)
    rel("p.v|same", o, oo)
    hh = L.post_attention_layernorm(x)
    g = L.gate_proj(hh); u = L.up_proj(hh)
    act = F.silu(g) * u
    gn, un = g.cpu().numpy(), u.cpu().numpy()
    actn = (gn / (np.float32(1) + np.exp(-gn))).astype(np.float32) * un
    rel("silu*up|same", act, actn)
    dn = L.down_proj(act)
    dno = O.linear_ptq(act.cpu().numpy(), sd["layers.0.down_proj.weight"], None, d)[0]
    rel("down|same", dn, dno)
    print("down_proj mode", L.down_proj._align_mode, getattr(L.down_proj, "_x_cap", None))
