#!/usr/bin/env python3
"""Trained-weights perplexity fixture: a small causal LM that is NOT noise.

Build container only (imports /root/reference).  For each family (OPT-style, Llama-style):
  1. the REFERENCE's own model class (`OPTQuantizedForCausalLM` / `LlamaQuantizedForCausalLM`) in bypass mode -- every
     quantiser switched off by its own `bypass` key -- is trained on CPU for a few hundred AdamW steps as a byte-level
     language model (4 layers, hidden 128, 2 heads of 64, 512 positions) on English text already in this image
     (/usr/lib/python3.10/pydoc_data/topics.py: the Python language reference as pydoc ships it);
  2. the weights are rounded to fp16 (what the fixture stores: <= 2 MB per family) and cast back to fp32;
  3. the reference's QUANTISED classes (W6A6 and W4A4 block_fp, configs/quantization/bfp_6bit.toml / bfp_4bit.toml [default])
     evaluate the perplexity of held-out text (/usr/share/common-licenses/GPL-3, verbatim-redistributable) over 16 chunks of
     512 tokens exactly as eval/eval_lm.py:41-63 does (batch 1: per-chunk loss * seq_len summed, ppl = exp(sum / tokens)),
     chunked as datasets/wikitext2.py:30-46 chunks its token stream (consecutive max_length windows, remainder dropped).
Writes tests/golden/trained.npz + trained.json: weights (fp16), token ids, per-chunk losses, loss, perplexity, the bypass
(un-quantised) perplexity and the parsed per-layer configs.  Data only.

    python tools/gen_trained_fixture.py            # ~10 minutes on 8 cores
"""
from __future__ import annotations

import json
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
from gen_golden_models import OUT, _jsonable, bfp_default, load_reference_models  # noqa: E402

TRAIN_TEXT = Path("/usr/lib/python3.10/pydoc_data/topics.py")
EVAL_TEXT = Path("/usr/share/common-licenses/GPL-3")
VOCAB, OFFSET = 260, 4                 # byte b -> token b + 4 (0..3: the specials the HF configs reserve)
T, CHUNKS = 512, 16
HIDDEN, LAYERS, HEADS = 128, 4, 2


def tokens_of(path: Path) -> np.ndarray:
    return np.frombuffer(path.read_bytes(), dtype=np.uint8).astype(np.int64) + OFFSET


def bypass_config():
    d = bfp_default(6, 6)
    d["bypass"] = True
    return {"default": d}


def build(family, mods, qcfg):
    opt, optc, llama, llamac = mods
    if family == "opt":
        cfg = optc.OPTQuantizedConfig(vocab_size=VOCAB, hidden_size=HIDDEN, num_hidden_layers=LAYERS, ffn_dim=4 * HIDDEN,
                                      max_position_embeddings=T, num_attention_heads=HEADS, dropout=0.0,
                                      word_embed_proj_dim=HIDDEN, quant_config=json.loads(json.dumps(qcfg)))
        return opt.OPTQuantizedForCausalLM(cfg), cfg
    cfg = llamac.LlamaQuantizedConfig(vocab_size=VOCAB, hidden_size=HIDDEN, intermediate_size=3 * HIDDEN,
                                      num_hidden_layers=LAYERS, num_attention_heads=HEADS, max_position_embeddings=T,
                                      quant_config=json.loads(json.dumps(qcfg)))
    return llama.LlamaQuantizedForCausalLM(cfg), cfg


def train(family, mods, steps, batch, seed):
    torch.manual_seed(seed)
    model, _ = build(family, mods, bypass_config())
    model.train()
    data = torch.from_numpy(tokens_of(TRAIN_TEXT))
    g = torch.Generator().manual_seed(seed + 1)
    opt_ = torch.optim.AdamW(model.parameters(), lr=3e-3, betas=(0.9, 0.95), weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt_, max_lr=3e-3, total_steps=steps, pct_start=0.05)
    t0 = time.time()
    for step in range(steps):
        starts = torch.randint(0, data.numel() - T - 1, (batch,), generator=g)
        ids = torch.stack([data[s:s + T] for s in starts.tolist()])
        loss = model(input_ids=ids, labels=ids).loss
        opt_.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt_.step()
        sched.step()
        if step % 50 == 0 or step == steps - 1:
            print(f"  {family} step {step:4d}  loss {float(loss):.4f}  ({time.time() - t0:.0f} s)", flush=True)
    # fp16-rounded weights are THE weights from here on
    sd = {k: v.detach().to(torch.float16) for k, v in model.state_dict().items()}
    return sd


@torch.no_grad()
def evaluate(family, mods, sd16, qcfg, chunks):
    """eval/eval_lm.py:41-63 with batch_size 1 over `chunks` [CHUNKS, T]"""
    model, cfg = build(family, mods, qcfg)
    missing = model.load_state_dict({k: v.to(torch.float32) for k, v in sd16.items()}, strict=True)
    model.eval()
    losses = []
    for c in chunks:
        ids = torch.from_numpy(c)[None]
        losses.append(float(model(input_ids=ids, labels=ids).loss))
    total = sum(l * 1 * T for l in losses)
    reduced = total / (T * len(losses))
    return losses, reduced, math.exp(reduced), cfg


@torch.no_grad()
def control(seeds=(1, 2, 3)):
    """The reference against ITSELF (--control; adds "control" to every evaluation of trained.json): the same quantised
    evaluation with every Linear output moved by one fp32 ulp up or down at random -- what a GEMM that sums in another order
    (another BLAS, another thread count, a GPU) does to it.  If the perplexity moves in the third decimal under that, then
    "identical perplexity to 3 d.p." is not a property fp32 arithmetic determines for this model, whatever computes it."""
    mods = load_reference_models()
    data = np.load(OUT / "trained.npz")
    meta = json.loads((OUT / "trained.json").read_text())
    chunks = data["input_ids"]
    for tag, m in meta.items():
        pre = tag + "/w/"
        sd16 = {k[len(pre):]: torch.from_numpy(data[k]) for k in data.files if k.startswith(pre)}
        for name in ("w6a6", "w4a4"):
            qcfg = m["evals"][name]["quant_config"]
            out = []
            for seed in seeds:
                model, _ = build(m["family"], mods, qcfg)
                model.load_state_dict({k: v.to(torch.float32) for k, v in sd16.items()}, strict=True)
                model.eval()
                g = torch.Generator().manual_seed(seed)

                def jitter(mod, inp, y):
                    up = torch.rand(y.shape, generator=g) < 0.5
                    return torch.where(up, torch.nextafter(y, torch.full_like(y, float("inf"))), torch.nextafter(y, torch.full_like(y, float("-inf"))))
                for mod in model.modules():
                    if isinstance(mod, torch.nn.Linear) and hasattr(mod, "x_quantizer"):
                        mod.register_forward_hook(jitter)
                losses = [float(model(input_ids=torch.from_numpy(c)[None], labels=torch.from_numpy(c)[None]).loss) for c in chunks]
                ppl = math.exp(sum(losses) / len(losses))
                ref = data[f"{tag}/{name}/chunk_losses"]
                out.append(dict(seed=seed, perplexity=ppl, d_perplexity=ppl - m["evals"][name]["perplexity"],
                                max_d_chunk_loss=float(np.abs(np.asarray(losses) - ref).max())))
                print(f"  {tag} {name} 1-ulp jitter seed {seed}: ppl {ppl:.5f} (d {out[-1]['d_perplexity']:+.2e}), max |d chunk loss| {out[-1]['max_d_chunk_loss']:.2e}", flush=True)
            m["evals"][name]["control"] = dict(what="reference vs itself: every quantised Linear's output moved by +-1 fp32 ulp at random", runs=out)
    (OUT / "trained.json").write_text(json.dumps(meta, indent=1))


def main():
    torch.set_num_threads(8)
    if "--control" in sys.argv:
        return control()
    mods = load_reference_models()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    ev = tokens_of(EVAL_TEXT)
    ev = ev[: (ev.size // T) * T].reshape(-1, T)                    # group_texts: consecutive windows, remainder dropped
    chunks = ev[:CHUNKS]
    arrays, meta = {"input_ids": chunks}, {}
    for family, seed in (("opt", 1000), ("llama", 2000)):
        print(f"training {family} ...", flush=True)
        sd16 = train(family, mods, steps, batch=8, seed=seed)
        tag = f"{family}_trained"
        for k, v in sd16.items():
            arrays[f"{tag}/w/{k}"] = v.numpy()
        m = dict(family=family, hidden_size=HIDDEN, num_layers=LAYERS, num_heads=HEADS, vocab_size=VOCAB, max_positions=T,
                 seq_len=T, num_chunks=CHUNKS, train_steps=steps, train_text=str(TRAIN_TEXT), eval_text=str(EVAL_TEXT), evals={})
        m.update(dict(ffn_dim=4 * HIDDEN) if family == "opt" else dict(intermediate_size=3 * HIDDEN))
        for name, qcfg in (("bypass", bypass_config()), ("w6a6", {"default": bfp_default(6, 6)}), ("w4a4", {"default": bfp_default(4, 4)})):
            t0 = time.time()
            losses, reduced, ppl, cfg = evaluate(family, mods, sd16, qcfg, chunks)
            arrays[f"{tag}/{name}/chunk_losses"] = np.asarray(losses, dtype=np.float64)
            m["evals"][name] = dict(loss=reduced, perplexity=ppl, quant_config=_jsonable(qcfg),
                                    parsed_quant_config=_jsonable({k: v for k, v in cfg.quant_config.items()}))
            if family == "llama":
                m["rms_eps"] = float(cfg.rms_norm_eps)
            print(f"  {tag} {name}: loss {reduced:.6f}  ppl {ppl:.4f}  ({time.time() - t0:.0f} s)", flush=True)
        meta[tag] = m
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / "trained.npz", **arrays)
    (OUT / "trained.json").write_text(json.dumps(meta, indent=1))
    print(f"trained fixtures: {sum(a.nbytes for a in arrays.values()) / 1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
