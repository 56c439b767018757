import sys, numpy as np, torch
sys.path.insert(0, "llm-mixed-q_amd"); sys.path.insert(0, "tests")
from mi355q import ops
import test_gpu_gemm as T
dev = torch.device("cuda:0")
x, _, _ = T._inputs(200, 8, 1024, 37, "outlier")
xt = torch.from_numpy(x).to(dev)
_, xm, xe = ops.block_fp_quantize(xt, 6, 8, 127, [1, 16], True, want_fake=False, want_packed=True, fast_zero_blocks=True)
ref = ops.bfp_align(xm, xe, 5, 127)
got = ops.block_fp_quantize_aligned(xt, 6, 8, 127)
torch.cuda.synchronize()
er, eg = ref.exp.cpu().numpy().reshape(200, 4, 16), got.exp.cpu().numpy().reshape(200, 4, 16)
bad = np.argwhere((er != eg).any(-1))
print("differing row-groups", len(bad), "counts", int(ref.sparse[0]), int(got.sparse[0]))
xe_ = xe.cpu().numpy().reshape(200, 4, 16); xm_ = np.abs(xm.cpu().numpy().reshape(200, 4, 16, 16)).max(-1)
for r, g in bad[:4]:
    print(r, g, "codes", xe_[r, g], "amax", xm_[r, g], "ref E", er[r, g, 0], "got E", eg[r, g, 0])
