"""Fake-quant kernels at Llama-7B shapes (BASELINE config 5): HBM GB/s at 8 B per element.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qbw -o q -- python tools/bench_quantizers.py run
    python tools/bench_quantizers.py report gpurun_out/qbw/q_kernel_trace.csv

`run` launches every (format, shape) REPS times in a fixed order; `report` reads the per-dispatch durations of the
quantiser kernels from the trace in that order and prints one JSON line per case."""
import csv, json, sys
sys.path.insert(0, 'llm-mixed-q_amd'); sys.path.insert(0, '.')
REPS = 10
SHAPES = [("act [2048,4096]", (2048, 4096), True), ("act [2048,11008]", (2048, 11008), True),
          ("probs [32,2048,2048]", (32, 2048, 2048), True), ("weight [4096,4096]", (4096, 4096), False),
          ("weight [11008,4096]", (11008, 4096), False)]
FORMATS = ["block_fp W6", "block_minifloat (8,4,8)", "block_log (8,8)"]
HBM_PEAK_GBS = 8000.0


def time_events():
    """no profiler: HIP events around REPS back-to-back calls (includes launch gaps: a lower bound on GB/s)"""
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    for name, shape, skip in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(*shape, generator=g) * (0.02 if not skip else 1.0)).to(dev)
        n = x.numel()
        for f in FORMATS:
            if f.startswith("block_fp"):
                fn = lambda: ops.block_fp_quantize(x, 6, 8, 127, [1, 16], skip)
            elif f.startswith("block_minifloat"):
                fn = lambda: ops.block_minifloat_quantize(x, 8, 4, 8, [1, 16], skip)
            else:
                fn = lambda: ops.block_log_quantize(x, 8, 8, [1, 16], skip)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(REPS):
                    fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / REPS * 1e-3)
            gbs = 8.0 * n / best / 1e9
            print(json.dumps({"kernel": f, "tensor": name, "elements": n, "bytes_per_element": 8, "us_per_call": round(best * 1e6, 2),
                              "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / HBM_PEAK_GBS, 3)}), flush=True)


def run():
    import torch
    from mi355q import ops
    dev = torch.device("cuda:0")
    for name, shape, skip in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(*shape, generator=g) * (0.02 if not skip else 1.0)).to(dev)
        for f in FORMATS:
            for _ in range(REPS):
                if f.startswith("block_fp"):
                    ops.block_fp_quantize(x, 6, 8, 127, [1, 16], skip)
                elif f.startswith("block_minifloat"):
                    ops.block_minifloat_quantize(x, 8, 4, 8, [1, 16], skip)
                else:
                    ops.block_log_quantize(x, 8, 8, [1, 16], skip)
            torch.cuda.synchronize()


def report(path):
    rows = [r for r in csv.DictReader(open(path)) if "quant_vec_kernel" in r["Kernel_Name"] or "quant_generic_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    it = iter(rows)
    for name, shape, skip in SHAPES:
        n = 1
        for d in shape:
            n *= d
        for f in FORMATS:
            d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in (next(it) for _ in range(REPS)))
            med = d[len(d) // 2] * 1e-9
            gbs = 8.0 * n / med / 1e9
            print(json.dumps({"kernel": f, "tensor": name, "elements": n, "bytes_per_element": 8, "median_us": round(med * 1e6, 2),
                              "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / HBM_PEAK_GBS, 3)}))


if __name__ == "__main__":
    {"run": run, "events": time_events}[sys.argv[1]]() if sys.argv[1] != "report" else report(sys.argv[2])
