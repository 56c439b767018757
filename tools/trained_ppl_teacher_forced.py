"""Perplexity on the trained-weights fixture, END TO END and TEACHER-FORCED side by side.

The north_star sentence is "identical Wikitext2 perplexity to 3 d.p.".  On the only trained weights this project can have
(tests/golden/trained.*: 4-layer byte-level OPT- and Llama-style LMs trained and evaluated with the REFERENCE's own classes,
eval/eval_lm.py:41-63, tools/gen_trained_fixture.py) the GPU harness differs from the reference by 1.2e-3 ... 2.4e-3 in five of
eight cases.  This tool puts next to that difference the figures that say where it comes from:

  * `oracle_d_perplexity`: the numpy oracle's forward (oracle/np_models.py, pinned to the reference's logits at 1e-6) against the
    reference's perplexity -- the restatement itself, float64 contractions;
  * `teacher_forced_d_perplexity`: every decoder LAYER of the GPU model is fed the oracle's (= the reference's) input to that
    layer instead of its own predecessor's output; the loss is formed from the last layer's output.  Differences cannot compound
    across layers any more: what is left is what ONE layer of HIP kernels (its Linears, its attention core, its norms) adds;
  * `teacher_forced_worst_layer_err`: the largest |GPU layer output - oracle layer output| / max |oracle layer output| over all
    layers and chunks on identical inputs;
  * `d_perplexity`: the GPU model end to end, as tests/test_gpu_model.py::test_perplexity_on_trained_weights measures it;
  * `reference_vs_itself_1ulp_jitter`: the fixture's control -- the reference against itself with every Linear output moved by
    one fp32 ulp.

    python tools/trained_ppl_teacher_forced.py [--chunks 16]      -> one JSON line per (model, config), also appended to
                                                                    gpurun_out/r06_trained_perplexity.jsonl
"""
import argparse, json, math, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "llm-mixed-q_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=16)
    a = ap.parse_args()
    from mi355q import harness as H
    from oracle import np_models as NM
    gold = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gold, "trained.json")))
    data = np.load(os.path.join(gold, "trained.npz"))
    dev = torch.device("cuda:0")
    chunks = data["input_ids"][: a.chunks]
    for tag in sorted(meta):
        m = meta[tag]
        pre = tag + "/w/"
        sd = {k[len(pre):]: data[k].astype(np.float32) for k in data.files if k.startswith(pre)}
        for name in ("w6a6", "w4a4"):
            ev = m["evals"][name]
            L = m["num_layers"]
            if m["family"] == "opt":
                cfg = H.TinyOPTConfig(vocab_size=m["vocab_size"], hidden_size=m["hidden_size"], ffn_dim=m["ffn_dim"], num_layers=L,
                                      num_heads=m["num_heads"], max_positions=m["max_positions"])
                qc = H.expand_quant_config(ev["quant_config"], L)
                model = H.TinyOPTForCausalLM(cfg, qc)
                oracle = lambda ids, taps: NM.opt_forward(sd, qc, ids, m["num_heads"], taps=taps)
            else:
                cfg = H.TinyLlamaConfig(vocab_size=m["vocab_size"], hidden_size=m["hidden_size"], intermediate_size=m["intermediate_size"],
                                        num_layers=L, num_heads=m["num_heads"], max_positions=m["max_positions"], rms_eps=m["rms_eps"])
                qc = H.expand_llama_quant_config(ev["quant_config"], L)
                model = H.TinyLlamaForCausalLM(cfg, qc)
                oracle = lambda ids, taps: NM.llama_forward(sd, qc, ids, m["num_heads"], m["rms_eps"], taps=taps)
            model.load_reference_state_dict(sd).to(dev).eval()
            e2e, tf, orc, worst = [], [], [], 0.0
            with torch.no_grad():
                model(torch.from_numpy(chunks[0])[None].to(dev))          # (first PTQ forward: quantises and packs the weights)
                for c in chunks:
                    ids = torch.from_numpy(c)[None].to(dev)
                    e2e.append(float(model(ids, labels=ids)[1]))
                    taps = {}
                    orc.append(oracle(c[None], taps)[1])
                    T = ids.shape[1]
                    mask = torch.full((T, T), torch.finfo(torch.float32).min, device=dev).triu(1)[None, None]
                    pos = torch.arange(T, device=dev)[None]
                    out = None
                    for i, layer in enumerate(model.layers):
                        xin = torch.from_numpy(np.ascontiguousarray(taps[f"hidden{i}"])).to(dev)
                        out = layer(xin, mask) if m["family"] == "opt" else layer(xin, mask, pos)
                        want = taps[f"hidden{i + 1}"]
                        worst = max(worst, float(np.abs(out.cpu().numpy() - want).max() / np.abs(want).max()))
                    norm = model.final_layer_norm if m["family"] == "opt" else model.norm
                    logits = model.lm_head(norm(out))
                    tf.append(float(torch.nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), ids[:, 1:].reshape(-1))))
            ppl = lambda ls: math.exp(sum(ls) / len(ls))
            ref_chunks = data[f"{tag}/{name}/chunk_losses"][: a.chunks]
            ref_ppl = ev["perplexity"] if a.chunks == 16 else ppl(list(ref_chunks))
            row = {"model": tag, "config": name, "chunks": len(chunks), "reference_perplexity": round(ref_ppl, 6),
                   "perplexity": round(ppl(e2e), 6), "d_perplexity": round(ppl(e2e) - ref_ppl, 6),
                   "same_to_3dp": round(ppl(e2e), 3) == round(ref_ppl, 3),
                   "teacher_forced_perplexity": round(ppl(tf), 6), "teacher_forced_d_perplexity": round(ppl(tf) - ref_ppl, 6),
                   "teacher_forced_same_to_3dp": round(ppl(tf), 3) == round(ref_ppl, 3),
                   "teacher_forced_worst_layer_err": float(f"{worst:.3e}"),
                   "oracle_perplexity": round(ppl(orc), 6), "oracle_d_perplexity": round(ppl(orc) - ref_ppl, 6),
                   "reference_vs_itself_1ulp_jitter": round(max(abs(r["d_perplexity"]) for r in ev["control"]["runs"]), 5),
                   "what": "teacher-forced: every decoder layer of the GPU model fed the oracle's (= the reference's) input to that "
                           "layer; the loss from the last layer's output"}
            print(json.dumps(row), flush=True)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                with open(os.path.join(ROOT, "gpurun_out", "r06_trained_perplexity.jsonl"), "a") as f:
                    f.write(json.dumps(row) + "\n")


if __name__ == "__main__":
    main()
