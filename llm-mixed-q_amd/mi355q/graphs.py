"""HIP-graph replay of a forward pass built from mi355q modules.

At small widths the kernels of a decoder layer are shorter than the Python / ctypes work that launches them (OPT-125m
width, 2048 tokens: ~400 us of kernels in ~490 us of wall time per layer); recorded once into a HIP graph the launches
cost the host nothing.  Everything the library does on the launch path is capturable: kernels on the current stream, buffers
cached per (shape, stream), no host reads while a stream is capturing (ops._capturing()).

    fwd = GraphedForward(lambda ids: model(ids)[0], (example_ids,))
    logits = fwd(ids)            # copies ids into the static input, replays, returns the static output

The first PTQ forward (weight quantisation + packing, the `auto` route decision: host reads) runs in the warm-up, outside the
graph.  Inputs must keep their shapes; outputs are overwritten by the next replay.
"""
from __future__ import annotations

import torch


def _tree_map(fn, obj):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree_map(fn, o) for o in obj)
    if isinstance(obj, dict):
        return {k: _tree_map(fn, v) for k, v in obj.items()}
    return obj


class GraphedForward:
    def __init__(self, fn, example_args, warmup: int = 3):
        if not all(isinstance(a, torch.Tensor) and a.is_cuda for a in example_args):
            raise RuntimeError("mi355q.graphs: inputs must be tensors on a HIP device")
        self.fn = fn
        self.static_in = [a.clone() for a in example_args]
        self.stream = torch.cuda.Stream(device=self.static_in[0].device)
        self.stream.wait_stream(torch.cuda.current_stream())
        # every library buffer created on this stream from here on is one the captured kernels keep a pointer to: the
        # per-stream caches in ops.py must never evict them
        from . import ops
        ops._StreamCache.pin_stream(self.static_in[0].device.index, self.stream.cuda_stream)
        # warm-up on the capture stream itself: the library's buffers (and the split-K workspace) are per stream
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(max(1, warmup)):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream), torch.no_grad():
            self.static_out = fn(*self.static_in)

    def __call__(self, *args):
        if len(args) != len(self.static_in):
            raise TypeError(f"expected {len(self.static_in)} inputs")
        for dst, src in zip(self.static_in, args):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise RuntimeError("mi355q.graphs: input shape / dtype differs from the captured one")
            dst.copy_(src)
        self.graph.replay()
        # the replay rewrote the outputs in place behind autograd's back: bump their version so that nothing keyed on
        # (data_ptr, _version) -- the quantised-activation reuse record of ops.py -- takes them for the previous replay's
        _tree_map(lambda t: (torch.autograd.graph.increment_version(t), t)[1], self.static_out)
        return self.static_out

    def clone_output(self):
        return _tree_map(lambda t: t.clone(), self.static_out)
