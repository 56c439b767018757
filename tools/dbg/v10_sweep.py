"""The small-tile GEMM (mi355q_gemm_v10.hip) against the launcher's current choice at the three places VERDICT r4 item 1 names:
(a) per-rank shard shapes, (b) under-filled single-GPU model shapes, (c) the 4096^3 headline.  GEMM alone, HIP events, us (TOPS).
    python tools/dbg/v10_sweep.py            # the table: columns = launcher | geometry / splits
    python tools/dbg/v10_sweep.py one        # one process under the current MI355Q_V10 / MI355Q_V8_SPLITS environment (JSON lines)
"""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
SHAPES = (("shard bench P=8", 4096, 512, 4096), ("shard bench P=4", 4096, 1024, 4096), ("shard bench P=2", 4096, 2048, 4096),
          ("shard OPT-1.3B q P=8", 2048, 256, 2048), ("shard OPT-1.3B fc1 P=8", 2048, 1024, 2048), ("shard Llama q P=8", 2048, 512, 4096),
          ("shard Llama up P=8", 2048, 1376, 4096),
          ("OPT-1.3B q_proj", 2048, 2048, 2048), ("OPT-1.3B fc1", 2048, 8192, 2048), ("Llama-7B v_proj", 2048, 4096, 4096),
          ("Llama-7B gate", 2048, 11008, 4096), ("headline", 4096, 4096, 4096))

BF16_SHAPES = (("OPT-1.3B fc2", 2048, 2048, 8192), ("OPT-1.3B fc2 P=8", 2048, 256, 8192), ("Llama-7B o_proj", 2048, 4096, 4096),
               ("Llama-7B down", 2048, 4096, 11008), ("Llama-7B down P=8", 2048, 512, 11008), ("OPT-125m fc2", 2048, 768, 3072),
               ("bench bf16 P=8", 4096, 512, 4096), ("bench bf16", 4096, 4096, 4096))


def one():
    sys.path.insert(0, os.path.join(root, "llm-mixed-q_amd")); sys.path.insert(0, root)
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    width = int(os.environ.get("SWEEP_WIDTH", "6"))

    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / n * 1e3
    # (clock ramp: some tens of ms of work before the first timing)
    z = torch.randn(4096, 4096, device=dev)
    for _ in range(30): z @ z
    if os.environ.get("SWEEP_BF16"):
        for name, M, N, K in BF16_SHAPES:
            x = torch.randn(M, K, device=dev) * torch.exp(torch.randn(M, 1, device=dev))
            w = torch.randn(N, K, device=dev) * 0.02
            xt = ops.block_fp_quantize_bf16_tiled(x, 6, 8, 127)
            wt = ops.block_fp_quantize_bf16_tiled(w, 6, 8, 127, reuse=False)
            y = torch.empty(M, N, device=dev)
            tg = t(lambda: ops.bf16_gemm_tiled(xt, wt, M, N, K, None, out=y))
            print(json.dumps({"shape": f"bf16 {name} {M}x{N}x{K}", "us": round(tg, 1), "TOPS": round(2.0 * M * N * K / tg / 1e6)}), flush=True)
        return
    for name, M, N, K in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
        _, wm, we = ops.block_fp_quantize(w, width, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
        wa = ops.bfp_align_rows(wm, we, width - 1, 127)
        y = torch.empty(M, N, device=dev)
        xa = ops.block_fp_quantize_aligned_rows(x, width, 8, 127)
        tg = t(lambda: ops.bfp_gemm_aligned(xa, wa, None, out=y))
        print(json.dumps({"shape": f"{name} {M}x{N}x{K}", "us": round(tg, 1), "TOPS": round(2.0 * M * N * K / tg / 1e6)}), flush=True)


def main():
    table, keys = {}, []
    # (geometry, ring depth, K splits); depth 0 = the launcher's rule
    combos = [(0, 0, 0)] + [(1, n, 1) for n in (3, 4, 5, 6)] + [(2, n, 1) for n in (3, 6)] + [(3, n, s) for n in (3, 4, 6, 8) for s in (1, 2)]
    if len(sys.argv) > 1:
        combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for g, n, s in combos:
        env = dict(os.environ)
        if g:
            env["MI355Q_V10"] = str(g)
            env["MI355Q_V8_SPLITS"] = str(s)
            if n:
                env["MI355Q_V10_NS"] = str(n)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True, cwd=root)
        key = "launcher" if not g else f"g{g}n{n}s{s}"
        keys.append(key)
        for l in r.stdout.splitlines():
            try:
                d = json.loads(l)
            except Exception:
                continue
            table.setdefault(d["shape"], {})[key] = d["us"]
        if r.returncode:
            print(f"[{key}] rc {r.returncode}: {r.stderr[-400:]}")
    print("us per GEMM; g1 = 128x256, g2 = 256x128, g3 = 128x128 tiles of mi355q_gemm_v10.hip, n = ring stages, s = K splits")
    print(" " * 44 + " ".join(k.rjust(8) for k in keys))
    for shape, row in table.items():
        print(f"{shape:44s}" + " ".join(f"{row.get(k, float('nan')):8.1f}" for k in keys))


if __name__ == "__main__":
    one() if len(sys.argv) > 1 and sys.argv[1] == "one" else main()
