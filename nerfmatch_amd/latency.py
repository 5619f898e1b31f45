"""Single-query latency of one localisation step: the reference's own operating point.

The reference's evaluation loop is batch 1 (nerfmatch_evaluator.py:631-724), it times `match_time` / `localize_time` per query
(:150-230, :502-629) and prints "Avg match time" (utils/metrics.py:589-593); SURVEY 8d metric (ii) is 1/t of ONE step.
`measure` times NeRFMatchEvaluator.eval_batch (lean render_novel_view + matcher forward, solver none, query2query) for batches of
Q queries with a synchronize on both sides of every step, and -- with a timing proxy around the C ABI -- the GPU time and the
number of native calls of a step.  Measurement aid (bench.py, scripts/perf_latency_q1.py): nothing in the product path imports it."""
import contextlib
import statistics
import time

import torch

from . import _lib

_HOST_ONLY = ("nm_nerf_pack",)  # C entry points that take pointers but launch nothing


class TimedLib:
    """Proxy of the ctypes handle: HIP events on torch's current stream around every native call that takes a stream.
    A call's span covers its kernels and the gaps between them; when the host is the bottleneck the span also holds the launch
    latency of its first kernel, so `gpu_ms` is an upper estimate of kernel time (the rocprofv3 kernel trace is the exact one)."""

    def __init__(self, real):
        self._real, self.spans = real, []

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        sig = _lib.SIGNATURES.get(name)
        if sig is None or sig[0] is not _lib.i32 or len(sig[1]) < 3 or sig[1][-1] is not _lib.vp or name.startswith(_HOST_ONLY):
            return fn

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*a)
            e1.record()
            self.spans.append((name, e0, e1))
            return rc

        return timed


@contextlib.contextmanager
def timed_lib():
    real = _lib.lib()
    proxy = TimedLib(real)
    _lib._lib = proxy
    try:
        yield proxy
    finally:
        _lib._lib = real


def measure(dev, renderer, H, W, kind="c2f", n=30, queries=1, warmup=5, gap_s=0.0, style=None):
    """-> dict(wall_ms median / p10 / p90 over n steps, gpu_ms = summed native-call spans of a step (median), native_calls per step,
    per_call {entry point: (calls per step, ms per step)}).  A step = eval_batch on one batch of `queries` queries.  gap_s: idle time
    between steps, outside the timed brackets (lets a kernel trace be cut into steps, scripts/latency_trace_summarize.py)."""
    from . import synth
    from .bench_match import build_evaluator

    ev, make_batch = build_evaluator(dev, H, W, queries=queries, kind=kind, style=style)
    if style == "peaked":  # thousands of matches per query (bench_match.CodedRenderer)
        from .bench_match import CodedRenderer

        renderer = CodedRenderer(renderer, ev.peaked_code)
    unnorm = synth.unnorm_scene()
    poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
    kw = dict(renderer=renderer, solver="none", query2query=True, mutual=True)

    def step(i):
        c2ws = torch.stack([poses[(i * queries + j) % 64] for j in range(queries)])
        return ev.eval_batch(make_batch(c2ws, unnorm), **kw)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if gap_s:
        time.sleep(gap_s)
    walls = []
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = step(warmup + i)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) * 1e3)
        if gap_s:
            time.sleep(gap_s)
    nmatch = int(out["num_matches"][0])
    # second pass with the timing proxy (its event records add host work: never mixed into the wall figure)
    gpu, calls, per = [], [], {}
    with timed_lib() as tl:
        for i in range(max(3, n // 3)):
            tl.spans = []
            step(warmup + n + i)
            torch.cuda.synchronize()
            tot = 0.0
            for name, e0, e1 in tl.spans:
                ms = e0.elapsed_time(e1)
                tot += ms
                c = per.setdefault(name, [0, 0.0])
                c[0] += 1
                c[1] += ms
            gpu.append(tot)
            calls.append(len(tl.spans))
    reps = len(gpu)
    series = list(walls)
    walls.sort()
    spec = dict(spec_batches=getattr(ev.model, "spec_batches", 0), spec_reruns=getattr(ev.model, "spec_reruns", 0), spec_cap=ev.model._spec_cap(1 << 30) if hasattr(ev.model, "_spec_cap") else None)
    return dict(**spec, wall_ms=statistics.median(walls), wall_ms_p10=walls[len(walls) // 10], wall_ms_p90=walls[(len(walls) * 9) // 10],
                gpu_ms=statistics.median(gpu), native_calls=statistics.median(calls), steps=n, queries=queries, matches=nmatch, series=series,
                per_call={k: (v[0] / reps, v[1] / reps) for k, v in per.items()})
