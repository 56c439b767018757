"""BASELINE config 4 as a torchrun-able model run: an OPT-1.3B-WIDTH decoder (hidden 2048, ffn 8192, 32 heads x 64; seeded
random weights -- no checkpoint exists here) under W4A4 block_fp with a mixed-precision [model_layer_i] section (the shape of
a search result of experiments/emnlp/configs/search/opt_1.3b_sst2.toml:24-52), B = 1, T = 2048, every quantised Linear
row-sharded over the ranks (mi355q.sharded.shard_model: BASELINE north_star; the reference places whole LAYERS instead,
cli/eval_perplexity.py:66-75).  Prints one JSON line per run: loss, ms per forward, collectives per forward.

    python tools/config4_sharded.py [--layers 24] [--tokens 2048] [--steps 5]                      # one GPU, unsharded
    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 tools/config4_sharded.py ...
    MI355Q_FORCE_DIST=1 python tools/config4_sharded.py ...      # world size 1 WITH the process group and its all-gathers
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llm-mixed-q_amd"))
import torch
import torch.distributed as dist


def quant_config(layers: int, knobs: bool, mx="auto"):
    """[default] W4A4 (bfp_4bit.toml / opt_1.3b_sst2.toml:39-52) + per-layer overrides of the kind the search writes"""
    d = dict(name="block_fp", bypass=False, is_ptq=True,
             data_in_width=4, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
             weight_width=4, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
             bias_width=4, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
    d["mi355q_mx"] = {"auto": "auto", "off": False, "on": True}[mx]      # (W4A4 on the MX scaled MFMA: DESIGN 5c)
    if knobs:
        d.update(mi355q_fused_attention=True, mi355q_grouped_linear=True, mi355q_fused_activation=True, mi355q_fused_norm=True,
                 mi355q_token_major_output=True, mi355q_fused_residual=True)
    cfg = {"default": d}
    # mixed precision: every third layer keeps its attention projections at 6 / 5 bits, every fourth its fc2 at 3-bit weights
    for i in range(layers):
        if i % 3 == 1:
            cfg[f"model_layer_{i}"] = {"self_attn": {"q_proj": dict(d, data_in_width=6, weight_width=5, bias_width=5),
                                                     "k_proj": dict(d, data_in_width=6, weight_width=5, bias_width=5)}}
        if i % 4 == 2:
            cfg.setdefault(f"model_layer_{i}", {})["fc2"] = dict(d, data_in_width=5, weight_width=3, bias_width=3)
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--tokens", type=int, default=2048)
    ap.add_argument("--mx", choices=["auto", "off", "on"], default="auto", help="config[\"mi355q_mx\"] of every layer")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--ffn", type=int, default=8192)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--vocab", type=int, default=50272)
    ap.add_argument("--no-knobs", action="store_true", help="the reference's step-by-step attention / separate launches")
    ap.add_argument("--gather", default="dense", choices=["dense", "segments", "quantised"])
    ap.add_argument("--no-heads", action="store_true", help="heads replicated, q / k / v gathered one by one (round 5's partition) "
                                                             "instead of the attention core on the rank's own heads")
    ap.add_argument("--vendor-head", action="store_true", help="lm_head on torch's fp32 GEMM (before round 6) instead of the split-bf16 product")
    a = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    force = world == 1 and os.environ.get("MI355Q_FORCE_DIST") == "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    from mi355q import harness, sharded
    torch.manual_seed(0)                                   # every rank builds the same full-precision model
    cfg = harness.TinyOPTConfig(vocab_size=a.vocab, hidden_size=a.hidden, ffn_dim=a.ffn, num_layers=a.layers, num_heads=a.heads,
                                max_positions=a.tokens)
    model = harness.TinyOPTForCausalLM(cfg, harness.expand_quant_config(quant_config(a.layers, not a.no_knobs, a.mx), a.layers))
    # (weights with per-channel spread, like trained ones: rows of different magnitude)
    with torch.no_grad():
        for p in model.parameters():
            if p.ndim == 2:
                p.mul_(torch.exp(0.5 * torch.randn(p.shape[0], 1)))
    model = model.to(dev).eval()
    model.mi355q_lm_head = "vendor" if a.vendor_head else "split"
    if world > 1 or force:
        sharded.shard_model(model, always_gather=force, gather=a.gather, heads=not a.no_heads)
    ids = torch.randint(0, a.vocab, (1, a.tokens), generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        for _ in range(2):                                  # first forward packs the weights, second settles the routes
            logits, loss = model(ids, labels=ids)
        torch.cuda.synchronize()
        sharded.COLLECTIVES.update(all_gather=0, bytes=0)
        from mi355q import ops as _ops
        _ops.vendor_gemm_calls(reset=True)
        if world > 1 or force:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            logits, loss = model(ids, labels=ids)
        torch.cuda.synchronize()
        if world > 1 or force:
            dist.barrier()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
    if rank == 0:
        print(json.dumps({"config": "BASELINE config 4: OPT-1.3B width, W4A4 block_fp + mixed [model_layer_i] sections",
                          "layers": a.layers, "tokens": a.tokens, "world": world, "forced_dist": force, "gather": a.gather, "head_sharded_attention": not a.no_heads,
                          "knobs": not a.no_knobs, "loss": round(float(loss), 6), "ms_per_forward": round(ms, 3),
                          "tokens_per_s": round(a.tokens / ms * 1e3, 1),
                          "all_gathers_per_forward": sharded.COLLECTIVES["all_gather"] // a.steps,
                          "gathered_MiB_per_forward": round(sharded.COLLECTIVES["bytes"] / a.steps / 2**20, 1),
                          "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                          "lm_head": "vendor fp32 GEMM" if a.vendor_head else "split-bf16 product",
                          # (torch's own GEMMs a forward called: none with the split-bf16 head; --vendor-head: the unquantised lm_head)
                          "vendor_gemm_calls_per_forward": {k: v // a.steps for k, v in sorted(_ops.vendor_gemm_calls().items())},
                          "logits_checksum": round(float(logits.double().abs().mean()), 8)}), flush=True)
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
