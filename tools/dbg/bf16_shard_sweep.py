"""the bf16 shard GEMM of the quantised gather (4096 x 512 x 4096, x in 8 column segments) under forced tile heights / K splits"""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(root, 'llm-mixed-q_amd'))
    import torch
    from mi355q import ops
    dev = torch.device('cuda:0'); M = K = 4096; P = 8; N = 512
    g = torch.Generator().manual_seed(0)
    y_full = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    wt = ops.block_fp_quantize_bf16_tiled(w, 6, 8, 127, reuse=False)
    segs = torch.stack([ops.block_fp_quantize_bf16_tiled(y_full[:, s * K // P:(s + 1) * K // P].contiguous(), 6, 8, 127, reuse=False).reshape(-1) for s in range(P)]).contiguous()
    plain = ops.block_fp_quantize_bf16_tiled(y_full, 6, 8, 127, reuse=False)
    out = torch.empty(M, N, device=dev)
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return round(a.elapsed_time(e) / n * 1e3, 1)
    print(json.dumps({"segmented_us": t(lambda: ops.bf16_gemm_tiled(segs, wt, M, N, K, None, out=out, segments=P)),
                      "plain_us": t(lambda: ops.bf16_gemm_tiled(plain, wt, M, N, K, None, out=out))}))
else:
    for rows in (0, 128, 256):
        for s in (0, 1, 2, 4):
            env = dict(os.environ, MI355Q_V8_TILE_ROWS=str(rows), MI355Q_V8_SPLITS=str(s))
            r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
            print(f"tile rows {rows:3d} splits {s}: {r.stdout.strip() or r.stderr[-200:]}")
