"""Upper-bound experiment for a deeper pipeline of the one-query loop (round 6): TWO host threads, each running the shipped two-stream loop
(render on XCDs 0-4 beside the previous query's matcher on XCDs 5-7) on its own evaluator, renderer and streams -- four streams in all, two
per partition.  Per-query time of the pair against one thread alone.

    python scripts/ab_loop_two_threads.py [queries per thread]
"""
import sys
import threading
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from bench import Batches
from nerfmatch_amd import _lib, synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]


def make():
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
    ren.to(dev).eval()
    ev, mk = build_evaluator(dev, H, W, queries=1)
    return ren, ev, mk


units = [make(), make()]
kw = dict(solver="none", query2query=True, mutual=True)
# a second set of partition streams for the second thread: the cache of _lib.partition_stream is keyed by (device, subset)
real_ps = _lib.partition_stream
tls = threading.local()


def ps(n_cus, first_cu=0, device=None, xcds=None):
    tag = getattr(tls, "tag", 0)
    if tag == 0:
        return real_ps(n_cus, first_cu, device, xcds)
    key = ("t", tag, n_cus, first_cu, xcds)
    st = _lib._PART_STREAMS.get(key)
    if st is None:
        saved = dict(_lib._PART_STREAMS)
        _lib._PART_STREAMS.clear()
        st = real_ps(n_cus, first_cu, device, xcds)  # a new stream with the same mask
        _lib._PART_STREAMS.update(saved)
        _lib._PART_STREAMS[key] = st
    return st


_lib.partition_stream = ps


def run(k, count, out):
    tls.tag = k
    torch.cuda.set_device(dev)
    ren, ev, mk = units[k]
    m = ev.eval_data_loader(data_loader=Batches(count, 6 + 7 * k, 1, poses, unnorm, mk), renderer=ren, **kw)
    torch.cuda.synchronize()
    out[k] = int(m["num_matches"].sum())


for k in (0, 1):  # warm-up, one after the other
    run(k, 6, {})
with _lib.steady_gc():
    o = {}
    t0 = time.perf_counter()
    run(0, n, o)
    one = (time.perf_counter() - t0) / n * 1e3
    print(f"one thread:  {one:.3f} ms per query ({n} queries)", flush=True)
    for rep in range(3):
        o = {}
        th = [threading.Thread(target=run, args=(k, n, o)) for k in (0, 1)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        two = (time.perf_counter() - t0) / (2 * n) * 1e3
        print(f"two threads: {two:.3f} ms per query ({2 * n} queries; matches {o})", flush=True)
