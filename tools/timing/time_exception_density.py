"""Row-scale int8 route vs per-block bf16 route of one Linear (2048 x 4096 -> 4096) as the activations carry more and
more blocks outside their row's exponent window: where the `auto` policy should change routes.  x = randn * row scale,
with a fraction p of the [1,16] blocks scaled by 2^-9 (far below any row window)."""
import json, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
import torch
import mi355q.quantize as Q
import mi355q.ops as ops
ops.REUSE_QUANTISED_INPUT = False
dev = torch.device("cuda:0")
def cfg(align):
    return dict(name="block_fp", is_ptq=True, bypass=False, data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127,
                data_in_block_size=[1, 16], weight_width=6, weight_exponent_width=8, weight_exponent_bias=127,
                weight_block_size=[1, 16], bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16],
                mi355q_align=align)
M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 4096, 4096)
torch.manual_seed(0)
fp = torch.nn.Linear(K, N, bias=True)
with torch.no_grad():
    fp.weight.normal_(0, 0.02)
for p in (0.0, 2e-4, 5e-4, 1e-3, 2e-3, 4e-3, 8e-3, 2e-2):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    blk = (torch.rand(M, K // 16, generator=g) < p)
    x = (x.view(M, K // 16, 16) * torch.where(blk, 2.0 ** -9, 1.0)[:, :, None]).view(M, K).to(dev)
    xa = ops.block_fp_quantize_aligned_rows(x, 6, 8, 127, bucket_cap=ops.ROW_BUCKET_CAP_MAX)
    over, fullest = ops.row_list_fill(xa.sparse, xa.rows, xa.list_cap)
    row = {"M": M, "K": K, "N": N, "p_exception_blocks": p, "fullest_x_bucket": fullest, "overflowed_rows": over}
    for align in ("rows", "rows_post", "blocks"):
        lin = Q.get_quantized_cls("linear", cfg(align)).from_float(fp, cfg(align)).to(dev)
        with torch.no_grad():
            for _ in range(5):
                lin(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                lin(x)
            b.record()
            torch.cuda.synchronize()
        row[align + "_us"] = round(a.elapsed_time(b) / 20 * 1e3, 1)
    print(json.dumps(row), flush=True)
