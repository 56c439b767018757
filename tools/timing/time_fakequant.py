import sys; sys.path.insert(0,'llm-mixed-q_amd'); sys.path.insert(0,'.')
import torch
from mi355q import ops
dev=torch.device('cuda:0')
def t(fn,n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e)/n*1e3
for shp in ((2048,4096),(2048,11008),(4096,4096)):
    for scale in (4.0, 1.0):
        x=torch.randn(*shp,device=dev)*scale
        print(shp, scale, 'bl %.1f us'%t(lambda: ops.block_log_quantize(x,8,8,[1,16],True)), 'bfp+mant %.1f'%t(lambda: ops.block_fp_quantize(x,6,8,127,[1,16],True,want_packed=True)), 'bfp %.1f'%t(lambda: ops.block_fp_quantize(x,6,8,127,[1,16],True)), 'bm+bias %.1f'%t(lambda: ops.block_minifloat_quantize(x,8,4,8,[1,16],True,want_bias=True)))
