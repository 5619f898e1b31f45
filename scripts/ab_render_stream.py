"""A/B of the two-stream pipeline of NeRFMatchEvaluator.eval_data_loader (round 6): per-query time of the loop at Q queries per batch with
the render (a) on the matcher's stream (the round-5 loop), (b) on a second plain stream, (c) on a compute-unit partition of n CUs with
the matcher unconfined, (d) both confined to complementary partitions -- and a bit-for-bit comparison of every query's match lists
against (a).

    python scripts/ab_render_stream.py [Q ...]        (default: 1 2 4 16)
"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from nerfmatch_amd import synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
KEEP = ("mpt2d_f", "mpt3d", "mconf", "mpt2d_c")


class Keep:
    """Indexable loader that keeps the batch dicts it handed out (the evaluator completes them in place)."""

    def __init__(self, n, Q, make_batch):
        self.n, self.Q, self.make_batch, self.batch_size, self.out = n, Q, make_batch, Q, {}

    def __len__(self):
        return self.n

    def __getitem__(self, b):
        q0 = b * self.Q
        d = self.make_batch(torch.stack([poses[(q0 + j) % 64] for j in range(self.Q)]), unnorm)
        self.out[b] = d
        return d


def run(ev, make_batch, Q, n, mode):
    ev.overlap_render = mode[0] != "off"
    ev.render_part = mode[1] if len(mode) > 1 else None
    ev.match_part = mode[2] if len(mode) > 2 else None
    ev.overlap_max_queries = 64
    ev.__dict__.pop('_x', None)
    kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
    torch.manual_seed(7)
    ev.eval_data_loader(data_loader=Keep(4, Q, make_batch), **kw)  # warm-up (stream creation, workspaces)
    torch.cuda.synchronize()
    torch.manual_seed(11)
    ld = Keep(n, Q, make_batch)
    t0 = time.perf_counter()
    m = ev.eval_data_loader(data_loader=ld, **kw)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / (n * Q) * 1e3
    outs = [{k: ld.out[b][k].clone() for k in KEEP} for b in range(n)]
    return wall, m["num_matches"].copy(), outs


def main():
    Qs = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 16]
    modes = [("off",), ("on", None), ("on", 224), ("on", 192), ("on", 160), ("on", 128), ("on", 192, 64), ("on", 160, 96), ("on", 128, 128),
             ("on", ("xcd", 0, 5), ("xcd", 5, 3)), ("on", ("xcd", 0, 6), ("xcd", 6, 2))]
    for Q in Qs:
        ev, make_batch = build_evaluator(dev, H, W, queries=Q)
        n = max(8, 64 // Q)
        base = None
        print(f"--- {Q} quer{'y' if Q == 1 else 'ies'} per batch, {n} batches per run", flush=True)
        for mode in modes:
            best = None
            for rep in range(3):
                wall, nm, outs = run(ev, make_batch, Q, n, mode)
                best = wall if best is None else min(best, wall)
            same = "(reference)"
            if base is None:
                base = (nm, outs)
            else:
                ok = bool((nm == base[0]).all()) and all(torch.equal(a[k], b[k]) for a, b in zip(outs, base[1]) for k in KEEP)
                same = "bit-identical" if ok else "DIFFERENT"
            tag = "one stream" if mode[0] == "off" else ("plain second stream" if mode[1] is None else
                                                        f"render on {mode[1]}" + (f", matcher on {mode[2]}" if len(mode) > 2 else ""))
            print(f"  {tag:38s} {best:7.3f} ms/query (best of 3)   matches {int(nm[0])}   {same}", flush=True)


def parts():
    """The two halves alone: the render of one query on a partition of n CUs, the matcher of one query on the complement, and both at once."""
    from nerfmatch_amd import _lib

    ev, make_batch = build_evaluator(dev, H, W, queries=1)
    batch = make_batch(torch.stack([poses[0]]), unnorm)
    ev._localize_begin(batch, ren, ev._opts(inerf_conf=None, iters=1, mutual=True, match_thres=0.0, solver="none", rthres=1, center_subpixel=False,
                                           query2query=True, retrieval_only=False, cached_pt=True, cache_iters=False, debug=False, match_oracle=False))
    torch.cuda.synchronize()
    K = batch["K"]

    def render(n=30):
        for _ in range(n):
            ren.render_novel_views((H, W), K[0], torch.stack([poses[1]]), unnorm, dev, downsample=8, want_im_pred=False)

    def match(n=30):
        for _ in range(n):
            ev.model.forward_finish(ev.model.forward_begin(batch, mutual=True))

    def timed(fn, st):
        with torch.cuda.stream(st):
            fn(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            fn(30)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 30 * 1e3

    print("--- the halves alone (ms per query, 30 back to back)")
    full = torch.cuda.Stream()
    print(f"  whole chip: render {timed(render, full):.3f}   matcher {timed(match, full):.3f}")
    for n in (224, 192, 160, 128, 96, 64):
        st = _lib.partition_stream(n, 0, dev)
        print(f"  {n:3d} CUs:    render {timed(render, st):.3f}   matcher {timed(match, st):.3f}", flush=True)
    for c in (6, 5, 3, 2):
        st = _lib.partition_stream(0, 0, dev, xcds=(0, c))
        print(f"  {c} XCDs:     render {timed(render, st):.3f}   matcher {timed(match, st):.3f}", flush=True)


def one(spec, Qs):
    """One setting in a process of its own (the mapping of streams to hardware queues depends on how many streams a process has made:
    two settings measured in one process are not independent).  spec: off | plain | <render>[,<match>] with N = a CU count, xA-B = XCDs A..B-1."""
    def part(t):
        return ("xcd", int(t[1:].split("-")[0]), int(t[1:].split("-")[1]) - int(t[1:].split("-")[0])) if t.startswith("x") else int(t)

    mode = ("off",) if spec == "off" else ("on", None) if spec == "plain" else ("on",) + tuple(part(t) for t in spec.split(","))
    for Q in Qs:
        ev, make_batch = build_evaluator(dev, H, W, queries=Q)
        n = max(8, 64 // Q)
        walls = [run(ev, make_batch, Q, n, mode)[0] for _ in range(4)]
        print(f"  {spec:14s} Q={Q:2d}  {min(walls):7.3f} ms/query (best of 4; all: {' '.join(f'{w:.3f}' for w in walls)})", flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["parts"]:
        parts()
    elif sys.argv[1:2] == ["one"]:
        one(sys.argv[2], [int(a) for a in sys.argv[3:]] or [1, 2, 4])
    else:
        main()
