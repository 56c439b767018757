"""In-kernel phase stamps of the tile GEMM (MI355Q_V8_STAMPS=1 selects the stamps build of the kernel): three stamped
launches on the clock ramp (an idle GPU), then -- `steady` argument -- 100 ms of un-stamped work and three more."""
import os, sys, time; sys.path.insert(0,'llm-mixed-q_amd'); sys.path.insert(0,'.')
import torch, bench
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev=torch.device('cuda:0')
x,w,b=bench.make_inputs(torch,dev,0)
_,wm,we=ops.block_fp_quantize(w,6,8,127,[1,16],False,want_fake=False,want_packed=True,fast_zero_blocks=True)
wa=ops.bfp_align_rows(wm,we,5,127); bq=ops.block_fp_quantize(b,6,8,127,[16],False)
y=torch.empty(4096,4096,device=dev)
def launches(n):
    for i in range(n):
        xa=ops.block_fp_quantize_aligned_rows(x,6,8,127); ops.bfp_gemm_aligned(xa,wa,bq,out=y); torch.cuda.synchronize(); print('---',flush=True)
if "steady" in sys.argv:
    # the ramp is driven through the bf16 flavour (no stamps, no printf): same tile kernel, same load
    xt=ops.block_fp_quantize_bf16_tiled(x,6,8,127); wt=ops.bf16_tile(w)
    t_end=time.time()+0.15
    while time.time()<t_end:
        for _ in range(10): ops.bf16_gemm_tiled(xt,wt,4096,4096,4096,bq)
        torch.cuda.synchronize()
    print('=== after 150 ms of work ===',flush=True)
launches(3)
