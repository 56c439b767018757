import torch
a=torch.randn(4,256,512,device='cuda').bfloat16(); b=torch.randn(4,512,128,device='cuda').bfloat16()
try:
    c=torch.bmm(a,b,out_dtype=torch.float32)
    ref=torch.bmm(a.float(),b.float())
    print('out_dtype ok', c.dtype, (c-ref).abs().max().item(), ref.abs().max().item())
except Exception as e:
    print('ERR', repr(e)[:300])
try:
    c=torch.matmul(a[0],b[0]) ; print(c.dtype)
    c=torch.mm(a[0],b[0],out_dtype=torch.float32); print('mm out_dtype ok',c.dtype)
except Exception as e:
    print('ERR2', repr(e)[:300])
