"""Is the tile GEMM's K loop limited by the issue schedule or by the chip's power management?  The same kernel on
operands of decreasing switching activity: benchmark data, constant small mantissas, zeros (in-kernel stamps)."""
import sys; sys.path.insert(0,'llm-mixed-q_amd'); sys.path.insert(0,'.')
import torch, bench
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev=torch.device('cuda:0')
x,w,b=bench.make_inputs(torch,dev,0)
for name,(xx,ww) in {"benchmark data":(x,w), "constant 1.0 / 0.02":(torch.ones_like(x), torch.full_like(w,0.02)), "zeros":(torch.zeros_like(x),torch.zeros_like(w))}.items():
    _,wm,we=ops.block_fp_quantize(ww,6,8,127,[1,16],False,want_fake=False,want_packed=True,fast_zero_blocks=True)
    wa=ops.bfp_align_rows(wm,we,5,127)
    y=torch.empty(4096,4096,device=dev)
    print("==",name,flush=True)
    for i in range(30):
        xa=ops.block_fp_quantize_aligned_rows(xx,6,8,127); ops.bfp_gemm_aligned(xa,wa,None,out=y)
    torch.cuda.synchronize()
