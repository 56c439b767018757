"""time the activation quantiser with a correction binding under MI355Q_CORR_DBG knock-outs (diagnostic)"""
import os, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tools/dbg')
import torch
from mi355q import ops
ops.CORR = True
import corr_check as cc
ops.REUSE_QUANTISED_INPUT = False
x, w, b = cc.inputs(4096, 4096, 4096)
wa, bq = cc.pack(w, b)
xt = x.to(cc.dev)
t0 = cc.timeit(lambda: ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127))
t1 = cc.timeit(lambda: ops.block_fp_quantize_aligned_rows(xt, 6, 8, 127, against=wa))
print(f"MI355Q_CORR_DBG={os.environ.get('MI355Q_CORR_DBG', '0'):>3s}: plain {t0:.2f} us, with binding {t1:.2f} us", flush=True)
