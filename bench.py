#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block-quantised linear path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 4096 x 4096 x 4096 block_fp W6A6,
block [1,16]; x = randn(seed 0) * exp(randn(M,1, seed 1)), w = randn(seed 2) * 0.02,
b = randn(seed 3) * 0.02, fp32, resident in HBM before the timed region.
One step = the steady-state PTQ LinearBlockFP forward (reference quantized_modules/linear.py:63-71
after the first call): dynamic activation quantise+pack -> int8-MFMA block GEMM against
pre-packed weights -> fp32 y (+bias).  FLOPs = 2*M*N*K per step.

N > 1 (one process per GPU, RCCL): the partition BASELINE.json's north_star names -- W split by rows
(out_features) over the ranks, x replicated, each rank computes y[:, shard], ONE all-gather of the fp32
shards into the rank-major [P, M, N/P] buffer, which the next layer's quantiser reads in place as P row segments
(mi355q/sharded.py ShardedRows, mi355q_block_fp_quantize_aligned_rows_seg: no permute copy) -- the same 4096^3 layer on
N GPUs: "scaling": "strong".  `--shard tokens` times independent replicas instead (each rank its own
4096 rows, no collective: weak scaling); the default N > 1 run reports that number too ("replicas").

`--workload quantizers` (BASELINE config 5): the memory-bound fake-quantisers (block_fp, block_minifloat,
block_log) at the Llama-7B activation / weight shapes, GB/s at 8 B per element against the HBM roofline.

After the timed region the step's output is checked against the oracle (64 sampled rows x all columns of y; the
activation quantiser's exponents and mantissas over the whole 4096 x 4096 tensor, bit for bit): exit code 1 on
mismatch.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# (the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks of one node
#  fails with hipIpcGetMemHandle: invalid argument; set before anything initialises HIP)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "llm-mixed-q_amd"))
sys.path.insert(0, str(ROOT))

M = N = K = 4096
CFG = dict(name="block_fp", is_ptq=True, bypass=False,
           data_in_width=6, data_in_exponent_width=8, data_in_exponent_bias=127, data_in_block_size=[1, 16],
           weight_width=6, weight_exponent_width=8, weight_exponent_bias=127, weight_block_size=[1, 16],
           bias_width=6, bias_exponent_width=8, bias_exponent_bias=127, bias_block_size=[16])
INT8_DENSE_PEAK_TFLOPS = 5000.0   # MI355X_MICROARCH.md: I8 MFMA = 2x the ~2.5 PF dense bf16 rate
HBM_PEAK_GBS = 8000.0             # HBM3E spec; ~6300 achievable by a copy (MI355X_MICROARCH.md)


def make_inputs(torch, device, rank):
    g = lambda s: torch.Generator().manual_seed(s)
    x = torch.randn(M, K, generator=g(0 + 1000 * rank)) * torch.exp(torch.randn(M, 1, generator=g(1 + 1000 * rank)))
    w = torch.randn(N, K, generator=g(2)) * 0.02
    b = torch.randn(N, generator=g(3)) * 0.02
    return x.to(device), w.to(device), b.to(device)


def cpu_baseline(torch):
    """The reference's CPU fake-quant path (torch-op-order port, oracle/torch_port.py) on this
    box's host cores, bounded to ~10-20 s: steady-state PTQ linear at the full 4096^3 size."""
    from oracle import torch_port as P
    cores = os.cpu_count() or 1
    x, w, b = make_inputs(torch, "cpu", 0)
    wq = P.block_fp_quantize(w, 6, 8, 127, [1, 16], False)
    bq = P.block_fp_quantize(b, 6, 8, 127, [16], False)
    # torch's elementwise ops stop scaling (and regress) far below the box's core count: try a few
    # thread counts once and time the fastest -- the fairest baseline this CPU gives
    best = None
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64, cores)}):
        torch.set_num_threads(nt)
        P.linear_ptq_step(x[:512], wq, bq, CFG)
        t0 = time.perf_counter()
        P.linear_ptq_step(x, wq, bq, CFG)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    first, nt = best
    torch.set_num_threads(nt)
    iters = max(2, min(10, int(10.0 / max(first, 1e-3))))
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        P.linear_ptq_step(x, wq, bq, CFG)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(2.0 * M * N * K / med / 1e12, 4), "unit": "TFLOP/s", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": round(med * 1e3, 2),
            "sample": f"full 4096x4096x4096 steady-state PTQ linear (fake-quant x + fp32 F.linear), median of {iters} iterations"}


def cpu_baseline_config5(torch):
    """BASELINE config 5's CPU side: the reference's block_minifloat / block_log fake-quantisers (torch-op-order ports,
    oracle/torch_port.py) on this box's host cores at [4096, 4096] fp32 -- what BASELINE.md section 2 quotes (870 / 185 ms
    per call on 8 cores in the survey container).  A few calls each, bounded to some seconds."""
    from oracle import torch_port as P
    cores = os.cpu_count() or 1
    torch.set_num_threads(min(cores, 32))
    x = torch.randn(4096, 4096, generator=torch.Generator().manual_seed(7)) * 4.0
    out = {"cores": torch.get_num_threads(), "kind": "port", "shape": "[4096, 4096] fp32, block [1,16]", "unit": "GB/s at 8 B per element"}
    for name, fn in (("block_minifloat_w8e4", lambda: P.block_minifloat_quantize(x, 8, 4, 8, [1, 16], True)),
                     ("block_log_w8", lambda: P.block_log_quantize(x, 8, 8, [1, 16], True))):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out[name] = {"ms": round(ts[1] * 1e3, 1), "GB/s": round(8.0 * x.numel() / ts[1] / 1e9, 2)}
    return out


def config3_summary(torch):
    """BASELINE config 3 as one object: the Llama-7B-SHAPED decoder at full depth (32 layers, 4096 / 11008, 32 heads x 128; seeded
    N(0, 0.02) weights -- no checkpoint here), W6A6 block_fp [1,16], B = 1, T = 2048, every knob on, eager and HIP-graph replay
    (tools/config3_full_depth.py; the reference's loop: eval/eval_lm.py:41-63).  No perplexity claim: random weights.
    Runs in a CHILD process with a timeout (a SIGALRM handler raises wherever the main thread happens to be -- inside a
    graph capture, inside a ctypes launch -- and a sticky HIP error or an out-of-memory there would take the headline's own
    verification down with it): whatever happens in there, this process only reads a JSON line or reports the failure."""
    import subprocess
    torch.cuda.empty_cache()
    # never takes the headline line down with it: needs ~35 GiB of a GPU that may be shared, and is bounded in time
    free, _total = torch.cuda.mem_get_info()
    if free < 48 * 2**30:
        return {"skipped": f"{free / 2**30:.0f} GiB free on the device; the 32-layer model needs 35"}
    cmd = [sys.executable, str(ROOT / "tools" / "config3_full_depth.py"), "--layers", "32", "--tokens", "2048", "--steps", "3",
           "--storage", "resident", "--no-parity", "--no-spread"]
    try:
        p = subprocess.run(cmd, cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        return {"error": "config3 took longer than 300 s (child process killed)"}
    except OSError as e:
        return {"error": repr(e)[:200]}
    line = next((l for l in reversed(p.stdout.splitlines()) if l.startswith("{")), None)
    if p.returncode != 0 or line is None:
        return {"error": f"child exit code {p.returncode}: {(p.stderr or p.stdout)[-200:]}"}
    try:
        r = json.loads(line)
    except ValueError as e:
        return {"error": repr(e)[:200]}
    keep = ("layers", "tokens", "loss", "ms_per_forward_eager", "tokens_per_s_eager", "ms_per_forward_graph", "tokens_per_s_graph",
            "graph_equals_eager", "resident_GiB_after_packing", "peak_GiB", "linear_routes", "vendor_gemm_calls_per_forward",
            "lm_head", "gated_mlp", "rotary_on_load", "attention_writes_o_proj_operand")
    out = {"what": "Llama-7B shape, 32 layers, W6A6 block_fp [1,16], B = 1, T = 2048, seeded random weights, every knob on "
                   "(child process: tools/config3_full_depth.py)"}
    out.update({k: r[k] for k in keep if k in r})
    return out


def robustness_summary(torch, ops, device, steps=50):
    """The headline step on operands that are NOT friendly to the row-scale int8 route: the same 4096^3
    W6A6 layer through the registry's LinearBlockFP with the default `auto` policy, activations with OUTLIER CHANNELS
    (tools/gen_golden.py's `outlier` style on top of the headline's row scales: K / 64 random channels x 60 -- the pattern the
    paper's figure 1 is about, README.md:11).  Reports the route the layer settles on, the density of exception blocks a
    row window of the int8 container would leave, the steady-state step time and the oracle check of its output."""
    import numpy as np
    import mi355q.quantize as Q
    from oracle import np_oracle as O
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    idx = torch.randint(0, K, (K // 64,), generator=g)
    x[:, idx] *= 60.0
    x = x.to(device)
    _, w, b = make_inputs(torch, "cpu", 0)
    cfg = dict(CFG, name="block_fp", is_ptq=True, bypass=False)
    fp = torch.nn.Linear(K, N, bias=True)
    with torch.no_grad():
        fp.weight.copy_(w)
        fp.bias.copy_(b)
    lin = Q.get_quantized_cls("linear", cfg).from_float(fp, cfg).to(device)
    out = {"what": "the same 4096^3 W6A6 step through LinearBlockFP (mi355q_align = auto) on activations with outlier channels: "
                   "K / 64 random channels x 60 on top of the headline's row scales"}
    try:
        xa = ops.block_fp_quantize_aligned_rows(x, CFG["data_in_width"], 8, 127, bucket_cap=ops.ROW_BUCKET_CAP_MAX)
        over, fullest = ops.row_list_fill(xa.sparse, xa.rows, xa.list_cap)
        words, nb = 8 + 8 * xa.list_cap, (M + 255) // 256
        nexc = int(xa.sparse[: 8 + nb * words].cpu()[8::words][:nb].sum())        # (entries RESERVED: rows that overflowed included)
    except Exception:                                         # noqa: BLE001
        over, fullest, nexc = None, None, None
    with torch.no_grad():
        for _ in range(6):                                    # (the policy reads its overflow word on a doubling schedule of calls)
            y = lin(x)
        ms = float("inf")
        for _ in range(3):                                    # (the best of three passes of `steps` calls: see config5_summary)
            torch.cuda.synchronize()
            a_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_ev.record()
            for _ in range(steps):
                y = lin(x)
            e_ev.record()
            torch.cuda.synchronize()
            ms = min(ms, a_ev.elapsed_time(e_ev) / steps)
    if lin._uses_bf16_route():
        route = "bf16 per-block exponents (tile GEMM on bf16 MFMA)"
    elif getattr(lin, "_mixed", None) is not None:
        cl = lin._mixed["classes"]
        route = (f"mixed contraction, one launch: {cl.K0} of {K} columns row-aligned on the int8 MFMA, {cl.K1} (the block columns with "
                 "outlier channels) as per-block bf16 on the bf16 MFMA")
    else:
        route = f"int8 {lin._align_mode}"
    pick = np.sort(np.random.default_rng(6).choice(M, size=32, replace=False))
    ref = O.bfp_linear_int(x.cpu().numpy()[pick], w.numpy(), b.numpy(), CFG)
    err = float(np.abs(y[torch.from_numpy(pick).to(device)].cpu().numpy() - ref).max() / np.abs(ref).max())
    out.update({"route": route, "rows_whose_exceptions_overflow_a_1016_entry_bucket": over, "fullest_x_bucket": fullest,
                "ms_per_step": round(ms, 4), "value": round(2.0 * M * N * K / ms / 1e9, 2), "unit": "TFLOP/s",
                "frac_of_int8_peak": round(2.0 * M * N * K / ms / 1e9 / INT8_DENSE_PEAK_TFLOPS, 4),
                "verify": {"ok": err <= 1e-5, "gemm_rows_checked": 32, "gemm_max_rel_err": err, "gemm_tol": 1e-5}})
    if nexc is not None:
        out["exception_blocks_of_a_row_window"] = nexc
        out["exception_block_density"] = round(nexc / (M * K / 16), 5)
    return out


def config5_summary(torch, ops, args, device, with_cpu):
    """BASELINE config 5 inside the default line: the `--workload quantizers` cases at a short step count, condensed
    (aggregate and slowest case, fractions of the 8 TB/s figure and of a device copy of the same tensor), with the CPU
    side beside them.  Runs outside the timed region of the headline metric."""
    class A:
        pass
    q = A()
    q.steps, q.warmup, q.clock_ramp_ms = 30, 5, 0.0            # (the GPU is at its working clocks already)
    q.passes = 3                                               # (each case: the best of three passes of 30 calls)
    r = quantizer_workload(torch, ops, q, device)
    worst = min(r["cases"], key=lambda c: c["GB/s"])
    per = {}
    for c in r["cases"]:
        d = per.setdefault(c["quantizer"], {"min_GB/s": 1e30, "max_GB/s": 0.0})
        d["min_GB/s"] = min(d["min_GB/s"], c["GB/s"]); d["max_GB/s"] = max(d["max_GB/s"], c["GB/s"])
    out = {"metric": r["metric"], "aggregate_GB/s": r["value"], "aggregate_frac_of_8TBs": round(r["value"] / HBM_PEAK_GBS, 3),
           "worst_case": {k: worst[k] for k in ("quantizer", "shape", "us", "GB/s", "frac_of_8TBs", "copy_GB/s", "frac_of_copy")},
           "per_quantizer": per, "cases": len(r["cases"]), "steps": q.steps, "passes": q.passes, "timing": "best of `passes` passes of `steps` calls per case",
           "shapes": "act[2048,4096], act[2048,11008], probs / causal_probs[32,2048,2048], w[4096,4096], w[11008,4096]"}
    if with_cpu:
        out["cpu_baseline"] = cpu_baseline_config5(torch)
    return out


def verify_capture(torch, ops, x, w, b, y, rows=64):
    """Right behind the timed step (nothing that runs later -- config5, robustness, config3 -- can make the headline's
    check fail or pass): what the oracle check needs, copied to the host -- the sampled rows of the step's y and of x, the
    operands, and the activation quantiser's exponents and mantissas over the whole 4096 x 4096 tensor."""
    import numpy as np
    pick = np.sort(np.random.default_rng(5).choice(x.shape[0], size=rows, replace=False))
    got = y[torch.from_numpy(pick).to(y.device)].cpu().numpy()
    _, xm, xe = ops.block_fp_quantize(x, CFG["data_in_width"], 8, 127, [1, 16], True, want_fake=False, want_packed=True)
    return {"pick": pick, "got": got, "xs": x.cpu().numpy(), "ws": w.cpu().numpy(), "bs": b.cpu().numpy(),
            "xm": xm.cpu().numpy(), "xe": xe.cpu().numpy(), "rows": rows}


def verify_check(cap):
    """Outside the timed region, host only: (1) the activation quantiser's shared exponents and mantissas at the FULL
    4096 x 4096 size, bit for bit against the oracle (BASELINE config 2); (2) `rows` sampled rows x all columns of
    the step's y against the oracle's exact integer contraction.  -> dict; "ok" False makes bench.py exit 1."""
    import numpy as np
    from oracle import np_oracle as O
    xs, ws, bs, pick, rows = cap["xs"], cap["ws"], cap["bs"], cap["pick"], cap["rows"]
    code = O.bfp_encode(xs, CFG["data_in_width"], 8, 127, [1, 16], True)
    exp_ok = bool(np.array_equal(cap["xe"].astype(np.int32) - 127, code.exp))
    mant_ok = bool(np.array_equal(cap["xm"].reshape(-1, 16).astype(np.int32), code.mant))
    ref = O.bfp_linear_int(xs[pick], ws, bs, CFG)
    err = float(np.abs(cap["got"] - ref).max() / np.abs(ref).max())
    tol = 1e-5            # north_star allows 1e-3 on the dequantised GEMM result; the int path is held to 1e-5
    return {"ok": exp_ok and mant_ok and err <= tol, "exponents_bit_exact": exp_ok, "mantissas_bit_exact": mant_ok,
            "blocks_checked": int(code.exp.size), "gemm_rows_checked": int(rows), "gemm_cols_checked": int(ref.shape[1]),
            "gemm_max_rel_err": err, "gemm_tol": tol, "captured": "right behind the timed step; checked on the host last"}


def measured_ceiling(torch, ops, device, xa, wa, bq, y):
    """the achievable figure next to the nominal one.  (1) `peak_measured`: back-to-back v_mfma_i32_16x16x64_i8 from
    registers on random int8 operands, every compute unit, two waves per SIMD, in the tile GEMM's register blocking, after a soak
    (csrc/mi355q_diag.hip) -- and the clock that loop ran at.  (2) `loop_clock_GHz`: the clock the tile GEMM's OWN K loop runs at
    (cdna guide, "DVFS give-back" item 6): two seconds of back-to-back launches of the kernel as it ships, then a few launches
    of its stamps build (delta s_memtime / delta s_memrealtime around the K loop, median over the workgroups; no stamp executes
    in the kernel the timed step runs), with that loop's cycles per K-step beside it."""
    import ctypes
    import numpy as np
    from mi355q import _lib
    lib = _lib.load_library()
    out = {}
    try:
        fn = lib.mi355q_debug_mfma_i8_peak
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                                 ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
        tops, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0)
        rc = fn(2000, 1500.0, ctypes.byref(tops), ctypes.byref(ghz), ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
        if rc == 0 and tops.value > 0:
            out["peak_measured"] = round(tops.value, 1)
            out["peak_measured_clock_GHz"] = round(ghz.value, 3)
            out["peak_measured_what"] = ("back-to-back v_mfma_i32_16x16x64_i8 from registers, random int8 operands, 256 workgroups x 8 "
                                         "waves (two per SIMD), 1.5 s soak, this device (csrc/mi355q_diag.hip)")
    except Exception as e:                                   # noqa: BLE001  (diagnostic: reported, never raised)
        out["peak_measured_error"] = f"{type(e).__name__}: {e}"[:200]
    try:
        hook = lib.mi355q_debug_v9_stamps
        hook.restype, hook.argtypes = None, [ctypes.c_void_p]
        tiles = ((xa.rows + 255) // 256) * ((wa.rows + 255) // 256)
        stamps = torch.zeros(tiles * 2 * 8, dtype=torch.int64, device=device)
        t_end = time.time() + 2.0
        while time.time() < t_end:
            for _ in range(50):
                ops.bfp_gemm_aligned(xa, wa, bq, out=y)
            torch.cuda.synchronize()
        hook(ctypes.c_void_p(stamps.data_ptr()))
        try:
            for _ in range(20):
                ops.bfp_gemm_aligned(xa, wa, bq, out=y)
            torch.cuda.synchronize()
        finally:
            hook(ctypes.c_void_p(0))
        sv = stamps.cpu().numpy().reshape(tiles, 2, 8).astype(np.int64)[:, 0, :]
        real = np.maximum(sv[:, 3] - sv[:, 2], 1)                  # (x 10 ns)
        ok = sv[:, 6] > 0
        if ok.any():
            ghz_wg = sv[ok, 6] / real[ok] * 0.1
            out["loop_clock_GHz"] = round(float(np.median(ghz_wg)), 3)
            out["loop_clock_GHz_min_max"] = [round(float(ghz_wg.min()), 3), round(float(ghz_wg.max()), 3)]
            out["loop_clk_per_kstep"] = round(float(np.median(sv[ok, 6])) / (K // 64), 1)
            out["loop_us"] = round(float(np.median(real[ok])) * 0.01, 2)
            out["loop_clock_what"] = ("stamps build of the same kernel after 2 s of back-to-back launches: delta s_memtime / delta "
                                      "s_memrealtime around the K loop, median over the workgroups; 1024 clk per K-step = the MFMA bound")
    except Exception as e:                                   # noqa: BLE001
        out["loop_clock_error"] = f"{type(e).__name__}: {e}"[:200]
    return out


def committed_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the COMMITTED PMC passes (counters cannot be read inside a timed
    run: one counter per rocprofv3 pass over tools/cdriver/step_driver, which runs the same step through the C ABI)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(ROOT / "profiles" / name) as f:
                kernels = json.load(f)["kernels"]
            # (round 6 added a fourth template argument to the tile kernel: both spellings name the same launch)
            for k in (kernel, kernel.replace("<1, false, false>", "<1, false, false, false>")):
                if k in kernels:
                    return int(kernels[k]["hbm_bytes_per_launch"]), name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def committed_profile_launch_us(kernel):
    """average duration of the dominant kernel in the COMMITTED rocprofv3 kernel stats of this same step (profiles/r06_bench_kernel_stats.csv,
    tools/prof/prof_step.sh): the figure the event-timed `avg_launch_ms` is to be read against"""
    import csv
    for name in ("r06_bench_kernel_stats.csv", "r05_bench_kernel_stats.csv"):
        try:
            with open(ROOT / "profiles" / name) as f:
                for row in csv.DictReader(f):
                    n = row.get("Name", "")
                    if n.startswith("void " + kernel.replace("<1, false, false>", "<1, false, false")) and "(" in n:
                        return float(row["AverageNs"]) / 1e3, int(row["Calls"]), name
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def clock_ramp(torch, step, ms):
    """Un-timed launches of the step until the GPU has left its idle clocks.  An MI355X that has been idle takes some tens of
    milliseconds of work to reach the clocks it then holds; 20 + 5 steps of this benchmark are 2.5 ms of work, so without
    this the whole timed region sits on the ramp (measured: 99.5 us per step against 92.0 with it, profiles/r02_clock_ramp.txt).
    Steady-state throughput is what the metric names.  Reported in the JSON line as `clock_ramp_ms`; --clock-ramp-ms 0 = off."""
    if ms <= 0:
        return
    # a COUNT of launches, not a deadline: with several ranks the step holds a collective, and every rank must issue the
    # same number of them (ten launches per requested millisecond: the step takes ~0.1 ms, the sharded one up to twice that)
    for i in range(max(1, int(ms * 10))):
        step()
        if i % 20 == 19:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def timed(torch, dist, world, device, step, steps, warmup):
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def quantizer_workload(torch, ops, args, device):
    """BASELINE config 5: fake-quantisers at the Llama-7B shapes, 8 B per element (fp32 in, fp32 out)."""
    shapes = {"act[2048,4096]": (2048, 4096), "act[2048,11008]": (2048, 11008), "probs[32,2048,2048]": (32, 2048, 2048),
              "causal_probs[32,2048,2048]": (32, 2048, 2048), "w[4096,4096]": (4096, 4096), "w[11008,4096]": (11008, 4096)}
    fns = {"block_fp_w6": lambda t, skip: ops.block_fp_quantize(t, 6, 8, 127, [1, 16], skip),
           "block_minifloat_w8e4": lambda t, skip: ops.block_minifloat_quantize(t, 8, 4, 8, [1, 16], skip),
           "block_log_w8": lambda t, skip: ops.block_log_quantize(t, 8, 8, [1, 16], skip)}
    rows, tot_bytes, tot_t = [], 0.0, 0.0
    for sname, shp in shapes.items():
        skip = not sname.startswith("w[")
        x = torch.randn(*shp, generator=torch.Generator().manual_seed(7)).to(device) * 4.0
        if sname.startswith("causal_probs"):
            # what the quantiser of the second attention product really reads: softmax rows under the causal mask -- half of
            # the [1,16] blocks exactly zero, which the exact zero-block rule rewrites in a second pass over the tensor
            x = torch.softmax(x + torch.full(shp[-2:], float("-inf"), device=device).triu(1), dim=-1)
        # the same 8 B per element as a plain device copy (torch's copy kernel): what this memory system gives a stream
        # of that size -- SURVEY 8(d) asks for the fraction of it next to the fraction of the 8 TB/s figure
        y = torch.empty_like(x)
        if not rows:                         # (first case: leave the idle clocks first, see clock_ramp)
            clock_ramp(torch, lambda: y.copy_(x), args.clock_ramp_ms * 5)      # (a 10-us copy: 50 per millisecond)
        for _ in range(args.warmup):
            y.copy_(x)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            y.copy_(x)
        e.record()
        torch.cuda.synchronize()
        copy_gbs = 8.0 * x.numel() / (a.elapsed_time(e) / args.steps * 1e-3) / 1e9
        del y
        for fname, fn in fns.items():
            for _ in range(args.warmup):
                fn(x, skip)
            ms = float("inf")
            for _ in range(getattr(args, "passes", 1)):        # (config5_summary: the best of three passes -- a box now and then stalls a
                torch.cuda.synchronize()                       #  stream for tens of ms, which inside a 0.5-ms pass would be the figure)
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.steps):
                    fn(x, skip)
                e.record()
                torch.cuda.synchronize()
                ms = min(ms, a.elapsed_time(e) / args.steps)
            gbs = 8.0 * x.numel() / (ms * 1e-3) / 1e9
            rows.append({"quantizer": fname, "shape": sname, "us": round(ms * 1e3, 2), "GB/s": round(gbs, 1),
                         "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 3), "copy_GB/s": round(copy_gbs, 1),
                         "frac_of_copy": round(gbs / copy_gbs, 3)})
            tot_bytes += 8.0 * x.numel()
            tot_t += ms * 1e-3
    worst = min(rows, key=lambda r: r["GB/s"])
    agg = tot_bytes / tot_t / 1e9
    return {"metric": "fake-quantiser HBM GB/s (Llama-7B shapes, block=16, 8 B/element)", "value": round(agg, 1), "unit": "GB/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "clock_ramp_ms": args.clock_ramp_ms,
            "ms_per_step": round(tot_t * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "block_fp W6 / block_minifloat (8,4,8) / block_log (8,8) fake-quantise, [1,16] blocks, "
                                   "Llama-7B activation, attention-probability and weight shapes, fp32 in -> fp32 out"},
            "roofline": {"bound": "hbm", "kernel": f"quant_vec_kernel ({worst['quantizer']} at {worst['shape']}: the slowest case)",
                         "achieved": worst["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(worst["GB/s"] / HBM_PEAK_GBS, 4),
                         "traffic": None, "copy_GB/s": worst["copy_GB/s"], "frac_of_copy": worst["frac_of_copy"]},
            "cases": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shard", choices=["out_features", "tokens"], default="out_features",
                    help="N > 1: the row-wise partition of W + all-gather (north_star), or independent replicas")
    ap.add_argument("--workload", choices=["gemm", "quantizers"], default="gemm")
    ap.add_argument("--clock-ramp-ms", type=float, default=60.0,
                    help="un-timed launches of the step for this long before the warm-up steps (GPU clock ramp; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed step's output")
    ap.add_argument("--no-config5", action="store_true", help="leave the fake-quantiser summary (BASELINE config 5) out of the line")
    ap.add_argument("--no-robustness", action="store_true", help="leave the outlier-channel variant of the step out of the line")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the measured int8-MFMA ceiling and the K loop's clock (roofline.peak_measured, loop_clock_GHz: ~4 s)")
    ap.add_argument("--no-config3", action="store_true", help="leave the full-depth Llama-7B-shape forward (BASELINE config 3) out of the line")
    ap.add_argument("--variant", type=int, default=0, help="GEMM kernel variant (0 = automatic)")
    ap.add_argument("--align", choices=["rows"], default="rows",
                    help="exponent alignment of the packed operands: whole rows (row-scale int8 GEMM); the 256-value-group "
                         "flavour was removed in round 5")
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: whatever native libraries print there (RCCL's "Librccl path" banner
    # arrives at process exit, behind the JSON line) goes to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from mi355q import ops
    # every timed step quantises its activation: the loop below feeds ONE tensor as a stand-in for a fresh activation per
    # step, which the module layer's shared-activation reuse (ops.REUSE_QUANTISED_INPUT) would recognise and skip
    ops.REUSE_QUANTISED_INPUT = False

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the block-quantised path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # (MI355Q_BENCH_FORCE_DIST=1: take the sharded code path -- process group, all-gather, layout fix-up -- at world size 1,
    #  so that a 1-GPU box can exercise it)
    force_dist = world == 1 and os.environ.get("MI355Q_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
    if args.variant:
        ops.set_gemm_variant(args.variant)

    if args.workload == "quantizers":
        out = quantizer_workload(torch, ops, args, device)
        if rank == 0:
            result_out.write(json.dumps(out) + "\n")
            result_out.flush()
        if world > 1:
            dist.destroy_process_group()
        return

    sharded = args.shard == "out_features" and (world > 1 or force_dist)
    xw, ww = CFG["data_in_width"], CFG["weight_width"]
    rows_mode = True
    quantize_x = ops.block_fp_quantize_aligned_rows

    def build(shard_w: bool, x_rank: int):
        """(step, x, w_local, b_local, y_local) of one mode; weight / bias packing is the one-off first PTQ forward of the
        reference and stays outside the timed region"""
        x, w, b = make_inputs(torch, device, x_rank)
        if shard_w:
            n_loc = N // world
            w, b = w[rank * n_loc:(rank + 1) * n_loc].contiguous(), b[rank * n_loc:(rank + 1) * n_loc].contiguous()
        _, wm, we = ops.block_fp_quantize(w, ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
        wa = ops.bfp_align_rows(wm, we, ww - 1, 127)
        bq = ops.block_fp_quantize(b, CFG["bias_width"], 8, 127, [16], False)
        n_out = w.shape[0]
        gathered = torch.empty(world, M, n_out, dtype=torch.float32, device=device) if shard_w else None
        # (sharded: the product is stored straight into this rank's segment of the gather buffer and the all-gather runs in place
        #  -- sendbuff == recvbuff + rank * count: the collective moves the OTHER ranks' segments only, no local copy)
        y = gathered[rank] if shard_w else torch.empty(M, n_out, dtype=torch.float32, device=device)

        def step():
            xa = quantize_x(x, xw, 8, 127)
            ops.bfp_gemm_aligned(xa, wa, bq, out=y)
            if shard_w:
                # rank-major [P, M, N/P]: what the sharded module hands on (sharded.ShardedRows); the next mi355q Linear's
                # quantiser reads the P row segments where they lie, so y [M, N] is never re-assembled
                dist.all_gather_into_tensor(gathered.view(world * M, n_out), y)
        return step, x, w, b, y

    step, x, w, b, y = build(sharded, 0 if sharded or world == 1 else rank)
    ops.gemm_timing(False)
    # the literal protocol first -- W warm-up steps, K timed steps, straight from the idle GPU (the clocks are still rising:
    # reported as `no_ramp`); then the clock ramp and the steady-state figure the line's `value` carries
    no_ramp = None
    if args.clock_ramp_ms > 0:
        dt0 = timed(torch, dist, world, device, step, args.steps, args.warmup)
        no_ramp = (dt0, None)
    clock_ramp(torch, step, args.clock_ramp_ms)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # Two passes of exactly K steps each, both bracketed as `timed` brackets them.  Pass 1 is the step as a caller runs it:
    # `value` and `ms_per_step`.  Pass 2 has the library record a HIP event pair around every launch of the dominant kernel on
    # the launch stream: `roofline.avg_launch_ms`.  One pass for both overstates the step: two event records per step are two
    # more packets in the queue between dependent kernels (measured: +5-6 us per 90-us step, reported as
    # `ms_per_step_with_events`).
    dt = timed(torch, dist, world, device, step, args.steps, 0)
    cap = verify_capture(torch, ops, x, w, b, y) if (rank == 0 and not args.no_verify) else None
    ops.gemm_timing(True)
    dt_events = timed(torch, dist, world, device, step, args.steps, 0)
    ops.gemm_timing(False)
    n_timed, gemm_avg_ms, gemm_min_ms = ops.gemm_timing_read()
    # ... and the dominant kernel alone, K launches back to back on the operands of the last step, no events: launch to
    # launch, i.e. its duration plus one dispatch gap -- what the event pair (and the profiler's per-dispatch signals) add to
    # a kernel's measured duration shows against this figure
    b2b_ms, ceiling = None, {}
    if not sharded and rows_mode:
        xa_last = quantize_x(x, xw, 8, 127)
        _, wm_b, we_b = ops.block_fp_quantize(w, ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True, fast_zero_blocks=True)
        wa_b, bq_b = ops.bfp_align_rows(wm_b, we_b, ww - 1, 127), ops.block_fp_quantize(b, CFG["bias_width"], 8, 127, [16], False)
        b2b_ms = timed(torch, dist, world, device, lambda: ops.bfp_gemm_aligned(xa_last, wa_b, bq_b, out=y), args.steps, args.warmup) / args.steps * 1e3
        if rank == 0 and not args.no_ceiling:
            ceiling = measured_ceiling(torch, ops, device, xa_last, wa_b, bq_b, y)
    n_out = w.shape[0]
    flops_kernel = 2.0 * M * n_out * K                        # one launch of the dominant kernel on this rank
    flops_job = 2.0 * M * N * K * (1 if sharded or world == 1 else world)
    value = flops_job * args.steps / dt / 1e12

    replicas = None
    if sharded:                                               # the independent-replica number next to it
        step_r, *_ = build(False, rank)
        dt_r = timed(torch, dist, world, device, step_r, args.steps, args.warmup)
        replicas = {"value": round(2.0 * M * N * K * world * args.steps / dt_r / 1e12, 2), "unit": "TFLOP/s", "scaling": "weak",
                    "ms_per_step": round(dt_r / args.steps * 1e3, 4),
                    "what": "each rank its own 4096 rows against replicated packed weights, no collective"}

    qgather = None
    if sharded and N % (32 * world) == 0 and N == K:
        # the same layer as one link of a CHAIN of sharded layers under gather = "quantised" (sharded.py): the x operand
        # arrives as the previous layer's gathered tiled-bf16 segments; this rank multiplies on the bf16 tile GEMM, quantises its
        # own output slice with the next layer's quantiser and the ranks all-gather THAT (2 bytes per value).  Reported beside the
        # headline step, never instead of it; any failure is reported, not raised.
        try:
            xq, wq, _ = make_inputs(torch, device, 0)
            n_loc = N // world
            wt_q = ops.block_fp_quantize_bf16_tiled((wq[rank * n_loc:(rank + 1) * n_loc] * 0.5).contiguous(), ww, 8, 127, reuse=False)
            segs = torch.stack([ops.block_fp_quantize_bf16_tiled(xq[:, r * n_loc:(r + 1) * n_loc].contiguous(), xw, 8, 127, reuse=False).reshape(-1)
                                for r in range(world)]).contiguous()
            y_q = torch.empty(M, n_loc, dtype=torch.float32, device=device)

            def step_q():
                ops.bf16_gemm_tiled(segs if world > 1 else segs[0], wt_q, M, n_loc, K, None, out=y_q, segments=world)
                mine = ops.block_fp_quantize_bf16_tiled(y_q, xw, 8, 127, out=segs[rank])      # (straight into this rank's segment:
                dist.all_gather_into_tensor(segs.view(-1), mine)          #  the collective in place; the chain feeds itself)
            dt_q = timed(torch, dist, world, device, step_q, args.steps, args.warmup)
            qgather = {"value": round(2.0 * M * N * K * args.steps / dt_q / 1e12, 2), "unit": "TFLOP/s", "scaling": "strong",
                       "ms_per_step": round(dt_q / args.steps * 1e3, 4), "gathered_MiB_per_rank": round(segs.numel() * (world - 1) / world / 2**20, 1),
                       "what": "one link of a chain of row-sharded layers with the NEXT layer's quantised operand gathered (tiled bf16, 2 bytes "
                               "per value; bf16 tile GEMM with x in column segments + quantiser on the rank's own slice + all-gather)"}
        except Exception as e:                                       # noqa: BLE001
            qgather = {"error": f"{type(e).__name__}: {e}"[:300]}

    failed = False
    if rank == 0:
        achieved = flops_kernel / (gemm_avg_ms * 1e-3) / 1e12
        # (the dominant kernel: the 256 x 256 tile GEMM of mi355q_gemm_v9.hip with its exception add-back -- round 4; MI355Q_V9_FIX=0
        #  puts the round-2 kernel bfp_gemm_v8<1, 8, 2, false> back)
        kname = "mi355q::bfp_gemm_v8<1, 8, 2, false>" if os.environ.get("MI355Q_V9_FIX") == "0" else "mi355q::bfp_gemm_v9<1, false, false>"
        traffic, tsrc = committed_traffic(kname) if rows_mode and not sharded else (None, None)
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "quantised-GEMM TFLOP/s (4096^2, block=16, W6A6 BFP)",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "clock_ramp_ms": args.clock_ramp_ms,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "int8",
            "data": "synthetic",
            "config": {"workload": "steady-state PTQ LinearBlockFP forward: x[4096,4096] fp32 -> fused quantise+pack+align (W6, block [1,16]) "
                                   "-> int8-MFMA block GEMM vs pre-packed W[4096,4096] (W6) + bias -> y fp32"
                                   + (f"; W split by out_features over {world} ranks, x replicated, RCCL all-gather of the fp32 shards "
                                      "into the rank-major buffer the next quantiser reads in place (no permute copy)" if sharded else ""),
                       "arithmetic": "int8 mantissa x int8 mantissa -> int32 (MFMA), fp32 row/block scaling, fp32 y",
                       "M_per_gpu": M, "N": N, "N_per_gpu": n_out, "K": K, "shard": args.shard if world > 1 else "none",
                       "align": args.align, "gemm_variant": ops.set_gemm_variant(args.variant)},
            "roofline": {"bound": "mfma", "kernel": kname.replace("mi355q::", "") + " (row-scale int8 tile GEMM)",
                         "achieved": round(achieved, 2), "peak": INT8_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / INT8_DENSE_PEAK_TFLOPS, 4),
                         "step_frac": round(value / world / INT8_DENSE_PEAK_TFLOPS, 4),
                         "step_frac_note": "the whole step (activation quantiser + GEMM + launch gaps) per GPU against the same peak",
                         "traffic": traffic,
                         "traffic_source": (f"committed profile profiles/{tsrc}: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE per launch, "
                                            "separate passes over tools/cdriver/step_driver (same step through the C ABI); not "
                                            "measured by this run") if traffic else None,
                         "avg_launch_ms": round(gemm_avg_ms, 4), "min_launch_ms": round(gemm_min_ms, 4), "launches_timed": n_timed,
                         "measured": f"second pass of {args.steps} steps with a HIP event pair on every launch of this kernel (library, launch "
                                     "stream; since round 6 the pair is attached to the kernel's own dispatch -- hipExtLaunchKernelGGL start / stop "
                                     "events: kernel start to kernel end -- instead of two marker records around it, which also timed the command "
                                     "processor's way from marker to dispatch to marker: 69-71 us on the same boxes); `value` is the first pass, "
                                     "without events",
                         "ms_per_step_with_events": round(dt_events / args.steps * 1e3, 4),
                         "back_to_back_launch_ms": None if b2b_ms is None else round(b2b_ms, 4),
                         "back_to_back_note": "the same kernel alone, launch to launch without events (duration + one dispatch gap)"},
        }
        prof_us, prof_calls, prof_src = committed_profile_launch_us(kname) if rows_mode and not sharded else (None, None, None)
        if prof_us:
            out["roofline"]["committed_profile"] = {
                "avg_launch_us": round(prof_us, 2), "launches": prof_calls, "frac": round(2.0 * M * n_out * K / (prof_us * 1e-6) / 1e12 / INT8_DENSE_PEAK_TFLOPS, 4),
                "source": f"profiles/{prof_src}: rocprofv3 --kernel-trace --stats of this step (tools/prof/prof_step.sh) on one box of round 6; not "
                          "measured by this run; `avg_launch_ms` is this run's event-timed figure for the same kernel"}
        if ceiling:
            # nominal AND achievable.  `frac` stays the fraction of the nominal 5 POPS; `frac_of_measured` prices the same
            # launch against what this device's int8 pipes deliver from registers on random operands
            out["roofline"].update(ceiling)
            if ceiling.get("peak_measured"):
                out["roofline"]["frac_of_measured"] = round(achieved / ceiling["peak_measured"], 4)
                if ceiling.get("loop_clock_GHz") and ceiling.get("peak_measured_clock_GHz"):
                    # ... and against that figure at the clock the K loop actually holds (power-limited: DESIGN 5a)
                    at_clock = ceiling["peak_measured"] * ceiling["loop_clock_GHz"] / ceiling["peak_measured_clock_GHz"]
                    out["roofline"]["peak_measured_at_loop_clock"] = round(at_clock, 1)
                    out["roofline"]["frac_of_measured_at_loop_clock"] = round(achieved / at_clock, 4)
        if no_ramp:
            jobf = 2.0 * M * N * K * (1 if sharded or world == 1 else world)
            out["no_ramp"] = {"value": round(jobf * args.steps / no_ramp[0] / 1e12, 2), "unit": "TFLOP/s",
                              "ms_per_step": round(no_ramp[0] / args.steps * 1e3, 4),
                              "step_frac": round(jobf * args.steps / no_ramp[0] / 1e12 / world / INT8_DENSE_PEAK_TFLOPS, 4),
                              "what": f"{args.warmup} warm-up + {args.steps} timed steps straight from an idle GPU, before the clock ramp"}
        if qgather:
            out["quantised_gather"] = qgather
        if replicas:
            out["replicas"] = replicas
        if rows_mode:
            # the one-off cost the steady-state step leaves out (reported separately): quantise + pack + row-align
            # this rank's weights and quantise its bias, as the first PTQ forward / pack_now() does; second of two runs
            ww = CFG["weight_width"]
            for _ in range(2):
                a_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_ev.record()
                _, wm_, we_ = ops.block_fp_quantize(w, ww, 8, 127, [1, 16], False, want_fake=False, want_packed=True,
                                                    fast_zero_blocks=True)
                ops.bfp_align_rows(wm_, we_, ww - 1, 127)
                ops.block_fp_quantize(b, CFG["bias_width"], 8, 127, [16], False)
                e_ev.record()
                torch.cuda.synchronize()
            out["weight_packing"] = {"ms": round(a_ev.elapsed_time(e_ev), 4), "once_per": "layer and checkpoint",
                                     "what": f"W[{w.shape[0]},{w.shape[1]}] fp32 -> W{ww} mantissas + exponents -> row-aligned "
                                             "tiled operand with its exception buckets, bias quantised; not in the timed step"}
        # (the GPU-side objects first, the host-side ones -- oracle check, CPU baselines: tens of seconds of all-core host work --
        #  last: the first GPU case measured right behind them ran 4-5x slow on this round's boxes, r5m / r5p / r5r, although neither
        #  an idle GPU nor busy host threads alone reproduce it, tools/dbg/host_after_cpu.py)
        if world == 1 and not args.no_config5:
            out["config5"] = config5_summary(torch, ops, args, device, False)
        if world == 1 and not args.no_robustness:
            try:
                out["robustness"] = robustness_summary(torch, ops, device)
            except Exception as e:                                   # noqa: BLE001  (reported, never raised: the headline stands)
                out["robustness"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_config3:
            out["config3"] = config3_summary(torch)
        if cap is not None:
            out["verify"] = verify_check(cap)
            failed = not out["verify"]["ok"]
        # (the outlier-channel variant's own oracle check counts too: ADVICE r5)
        rob = out.get("robustness", {})
        if isinstance(rob.get("verify"), dict) and not rob["verify"].get("ok", True):
            failed = True
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(torch)
            if "config5" in out:
                out["config5"]["cpu_baseline"] = cpu_baseline_config5(torch)
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: the timed step's output (or the robustness variant's) does not match the oracle")


if __name__ == "__main__":
    main()
