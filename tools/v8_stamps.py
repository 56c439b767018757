import sys; sys.path.insert(0,'llm-mixed-q_amd'); sys.path.insert(0,'.')
import torch, bench
from mi355q import ops
import mi355q.ops as _ops_mod; _ops_mod.REUSE_QUANTISED_INPUT = False   # (the loop below re-quantises ONE tensor on purpose)
dev=torch.device('cuda:0')
x,w,b=bench.make_inputs(torch,dev,0)
_,wm,we=ops.block_fp_quantize(w,6,8,127,[1,16],False,want_fake=False,want_packed=True,fast_zero_blocks=True)
wa=ops.bfp_align_rows(wm,we,5,127); bq=ops.block_fp_quantize(b,6,8,127,[16],False)
y=torch.empty(4096,4096,device=dev)
for i in range(3):
    xa=ops.block_fp_quantize_aligned_rows(x,6,8,127); ops.bfp_gemm_aligned(xa,wa,bq,out=y); torch.cuda.synchronize(); print('---',flush=True)
