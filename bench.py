#!/usr/bin/env python3
"""Benchmark of the NeRFMatch hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 (or NM_FORCE_DIST=1) and no WORLD_SIZE in the environment starts the ranks itself:
it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process
before anything touches the GPU, relays the child's JSON line as its last stdout line and exits with the child's code.

One "step" = one batch of Q query images on every rank (--queries, default 16; the reference's eval loop takes the
batch size as an argument, nerfmatch_evaluator.py:726-731,864-869, default 1).  Two timed regions of EXACTLY K steps
each (SURVEY.md section 8d defines two metrics):
  A  NerfRenderer.render_novel_views of Q 640x480 queries at downsample 8: Q x 4800 rays x (S+S) samples through the
     coarse and fine NeRF, EVERY sample evaluated and every output the reference's render_rays returns
     -> `value` = rays*samples/sec (whole job);
  B  NeRFMatchEvaluator.eval_data_loader (the reference's localisation driver; solver "none", query2query) over K
     batches of Q queries per rank: the lean render (pt3d / pt_feat only, as the evaluator's loop reads them; both passes on
     `--precision`) followed by the coarse-to-fine 2D-3D match against the rendered points (image backbone and PnP
     excluded: third party) -> `query_images_per_sec`.
Queries shard over ranks with no data-path collective; the per-query pose-candidate records are all-gathered once at
the end of each region (RCCL over xGMI), inside the timed region.

Extra legs, each an object of its own in the line (never `value`):
  variants.zero_tail_skip  region A with the fine pass skipping the zero-width intervals the reference's resampler leaves
                      (identical outputs) -- round 1/2's headline definition;
  variants.coarse_fp16x1   region B with the coarse pass of the lean render on ONE fp16 product (opt-in, narrower arithmetic);
  variants.reference_geometry  regions A and B at the reference's shipped geometry: 480x480 -> 3600 rays / tokens, 128+128 samples
                      (configs/nerfmatch/nerfmatch_7scenes_sfm_c2f.yaml:12, configs/nerf/nerf_7scenes_mip_sfm.yaml:30,38);
  variants.single_product  BASELINE config 3 (16-bit operands, one product per block): the whole render on the fp16x1 kernel, error stated;
  variants.attention_fp8   BASELINE config 5: Cambridge NeRF, 256+256 samples, matcher attention on fp8 MFMA, error stated;
  variants.cambridge  region A with the Cambridge NeRF (appearance embedding 16, white background: BASELINE configs 4/5);
  variants.cambridge_s256  the same at 256 + 256 samples per ray (config 5's ray length);
  variants.inerf_step_ms / multi_pair_k3_ms / train_step_ms / cache_frames_per_s   the SURVEY 8f rows (iNeRF refinement, multi-pair forward, one
                      training step of the matcher head, scene-feature cache writer), a few driver-timed steps each;
  mini                the coarse-only model's 4800 x 4800 dual-softmax + mutual NN (BASELINE config 2), MFMA roofline of the fused matching;
  roofline_b          region B's second kernel (attn32_v3_kernel), same definition as `roofline`.
`--samples 128|256` runs everything at that sample count (the shipped yaml value is 128; config 5 asks for 256);
`--hw 480x480` runs everything at that image size.

Prints ONE JSON line on rank 0 (see the task contract): metric rays*samples/sec (whole job), plus
  roofline     : dominant kernel, FLOP/launch / mean launch duration measured with HIP events on the launch stream
                 around the launches of the K TIMED steps (warm-up launches excluded), against the MFMA peak;
  cpu_baseline : the oracle (CPU restatement of the reference, torch-CPU fp32) timed on the host, one full 4800-ray query
                 (rank 0, N=1 only).
"""
import argparse
import contextlib
import json
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_SAMPLE_PASS = {"7scenes": 1_214_464, "cambridge": 1_218_560}  # 2 x MAC per sample and pass: SURVEY.md section 8d
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "fp16x3": 2500.0}  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / dense bf16 = fp16 MFMA peaks
PEAK_HBM_GBS = 8000.0
KERNEL = {"fp32": "nerf_fwd_kernel", "bf16x3": "nerf_fwd_bf16x3_kernel", "fp16x3": "nerf_fwd_fp16x3_kernel"}
# MFMA FLOPs the split kernels (fp16x3 / bf16x3) EXECUTE per sample and pass: 3 products per fp32 product, K padded 90->96 / 27+16->48
BF16X3_EXEC_FLOP_PER_SAMPLE_PASS = 3 * 2 * (96 * 256 + 4 * 65536 + (96 + 256) * 256 + 2 * 65536 + (256 + 48) * 128)  # (no feature_linear: folded into the views layer at pack time)
H, W, DS = 480, 640, 8  # BASELINE.json: synthetic 640x480 queries (--hw overrides)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=64, help="samples per ray per pass (coarse and fine)")
    ap.add_argument("--hw", default=f"{H}x{W}", help="query image size HxW (default 480x640 = BASELINE's 640x480 queries; the reference's yamls use 480x480)")
    ap.add_argument("--queries", type=int, default=16, help="query images per step per GPU (the reference's eval batch_size; 1 = its default)")
    ap.add_argument("--precision", choices=["fp16x3", "bf16x3", "fp32"], default="fp16x3",
                    help="matrix-core arithmetic of the fused NeRF kernel: fp16x3 = fp16 MFMA on hi/lo-split fp32 operands (default; 22 mantissa "
                         "bits, fp32-class results also on trained-like scenes), bf16x3 = the same split with bf16 parts (16 bits), "
                         "fp32 = v_mfma_f32_32x32x2_f32; the matcher's contractions run on bf16x3 unless fp32 is chosen")
    ap.add_argument("--variant", choices=["7scenes", "cambridge"], default="7scenes",
                    help="NeRF of regions A/B: 7scenes (BASELINE config 3, the headline) or cambridge (appearance embedding, white bg)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the other-precision / full-evaluation / cambridge / mini legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="render only")
    return ap.parse_args()


def cpu_baseline(S, seed_sd):
    """Oracle render of ONE full query (4800 rays x (S+S) samples).  The thread count is chosen by a short sweep (more
    threads than physical cores available to the container only slows torch-CPU down); the reported value is the median
    of 3 runs after 1 warm-up at the best setting (about 20 s of CPU work on the GPU box's host)."""
    from nerfmatch_amd import synth
    from oracle import nerf_oracle as no

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    K = synth.intrinsics(H, W)
    rays = no.make_rays(H, W, K, synth.camera_pose(1), ds=DS).contiguous()
    R = rays.shape[0]
    t_rand, jit = synth.uniform01((R, S + 1), 1), synth.resample_jitter((R, S + 1), 2)

    def run(r, tr, jt):
        t0 = time.perf_counter()
        no.render_rays(seed_sd, r, tr, jt, S, S, stop_layer=3)
        return time.perf_counter() - t0

    best_n, best_t = 1, float("inf")
    for n in sorted({c for c in (4, 8, 16, 32, 64, 128) if c <= avail} | {min(avail, 8)}):
        torch.set_num_threads(n)
        run(rays[:150], t_rand[:150], jit[:150])
        dt = run(rays[:600], t_rand[:600], jit[:600])
        if dt < best_t:
            best_n, best_t = n, dt
        if dt > 2.0 * best_t:
            break
    torch.set_num_threads(best_n)
    times = [run(rays, t_rand, jit) for _ in range(4)]
    med = statistics.median(times[1:])
    return dict(value=R * 2 * S / med, unit="rays*samples/s", cores=best_n, kind="port",
                sample=f"oracle.render_rays on one full query: {R} rays x ({S}+{S}) samples, median of 3 runs after 1 warm-up, "
                       f"torch-CPU fp32, {best_n} threads (best of a sweep; {avail} logical CPUs visible)")


class Batches:
    """Indexable stand-in for a DataLoader: batch b of the GLOBAL sequence holds queries b*Q .. b*Q+Q-1 (poses cycle through
    64 synthetic cameras).  The evaluator deals the batches round-robin over ranks."""

    def __init__(self, n, first, Q, poses, unnorm, make_batch):
        self.n, self.first, self.Q, self.poses, self.unnorm, self.make_batch = n, first, Q, poses, unnorm, make_batch
        self.batch_size = Q

    def __len__(self):
        return self.n

    def __getitem__(self, b):
        q0 = (self.first + b) * self.Q
        return self.make_batch(torch.stack([self.poses[(q0 + j) % 64] for j in range(self.Q)]), self.unnorm)


class KernelProbe:
    """ops.KERNEL_PROBE: HIP events on the launch stream around single native calls, switched on for the timed regions only."""

    def __init__(self):
        self.on, self.events = False, {}

    def __call__(self, tag, flop):
        return self._ctx(tag, flop) if self.on else contextlib.nullcontext()

    @contextlib.contextmanager
    def _ctx(self, tag, flop):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        yield
        e1.record()
        self.events.setdefault(tag, []).append((e0, e1, flop))

    def take(self, tag, min_flop=0.0):
        """-> (calls, mean ms per call, sum of flop, sum of seconds) of the recorded calls of `tag` with flop >= min_flop (call after a synchronize)."""
        ev = [(a.elapsed_time(b), f) for a, b, f in self.events.pop(tag, []) if f >= min_flop]
        if not ev:
            return None
        tot_ms = sum(m for m, _ in ev)
        return len(ev), tot_ms / len(ev), sum(f for _, f in ev), tot_ms * 1e-3


def mfma_probe(dev):
    """TFLOP/s of a bare 16-bit MFMA stream on every SIMD of this GPU (nm_probe_mfma_f16): what the chip sustains at its power limit."""
    import ctypes as C
    import torch
    from nerfmatch_amd import _lib, ops
    try:
        L = _lib.lib()
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        wgs, rounds = cus, 20000
        sink = torch.empty(wgs * 256, device=dev, dtype=torch.float32)
        call = lambda: _lib.check(L.nm_probe_mfma_f16(C.c_void_p(sink.data_ptr()), wgs, rounds, ops.stream()), "nm_probe_mfma_f16")
        call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            call()
        e1.record()
        torch.cuda.synchronize()
        return wgs * 4 * rounds * 24 * 32768.0 / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e12
    except Exception:  # a measurement aid: never fails the bench line
        return None


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD `torch.distributed.run` (never os.exec*; this
    process has only imported torch, no GPU call yet), relay its output, print its JSON line last, exit with its code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in child.stdout:
        if ln.startswith('{"metric"'):
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    raise SystemExit(rc if rc else (0 if line is not None else 1))


def pmc_traffic(names, applicable=True):
    """Sum of `derived.traffic_bytes` of the named profiles/*.json (measured per launch at the workload their header states), or None."""
    if not applicable:
        return None
    tot = 0.0
    for n_ in names:
        f_ = ROOT / "profiles" / n_
        if not f_.exists():
            return None
        tot += json.load(open(f_))["derived"]["traffic_bytes"]
    return tot


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("NM_FORCE_DIST") == "1"):
        self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    global H, W
    H, W = (int(v) for v in args.hw.lower().split("x"))
    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # NM_BENCH_SHARE_GPU=1 (tests only: the line says so) puts every rank on cuda:0 and the collectives on gloo -- RCCL refuses two ranks
    # on one device --, so that the N > 1 logic of this file (dealing of the batches, gathered records, MAX over ranks, rank 0's line) runs on
    # the one GPU a test box has
    share = os.environ.get("NM_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NM_FORCE_DIST=1 exercises the RCCL code path (init, barrier, all_gather, all_reduce) even at world size 1
    use_dist = world > 1 or os.environ.get("NM_FORCE_DIST") == "1"
    if use_dist:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import nerfmatch_amd
    from nerfmatch_amd import synth, ops
    from nerfmatch_amd._lib import steady_gc
    from nerfmatch_amd.nerf.renderer import NerfRenderer
    import nerfmatch_amd.nerf.renderer as rmod

    S, Q, Ksteps, Wsteps = args.samples, args.queries, args.steps, args.warmup
    R = (H // DS) * (W // DS)
    unnorm = synth.unnorm_scene()
    poses = [unnorm @ synth.camera_pose(seed=s_) for s_ in range(64)]

    def make_renderer(variant, samples=None, style=None):
        app = variant == "cambridge"
        r_ = NerfRenderer(synth.nerf_config(variant, num_pts=samples or S), num_frames=8 if app else None, training=False, stop_layer=3)
        sd_ = synth.nerf_state_dict(seed=0, app_vocab=8 if app else 0, density_bias=0.0 if style else 3.0, style=style)
        r_.load_state_dict(sd_)
        r_.to(dev).eval()
        r_.precision = args.precision
        return r_, sd_

    ren, sd = make_renderer(args.variant)

    # ---- instrumentation of the dominant kernel: HIP events around every nm_nerf_fwd launch of the TIMED steps, recorded
    # on the stream the launches go to (torch's current stream)
    raw_fwd = ops.nerf_fwd
    rec = dict(on=False, events=[])
    kprobe = KernelProbe()
    ops.KERNEL_PROBE = kprobe

    def timed_fwd(*a, **kw):
        if not rec["on"]:
            return raw_fwd(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = raw_fwd(*a, **kw)
        e1.record()
        # samples the launch really evaluates: all R*S, or R*(S/2+1) when the bf16x3 kernel skips the zero-width tail of
        # the fine pass (NM_NERF_ZERO_TAIL; the skipped samples have weight exactly 0 in every output)
        rr, ss = a[2].shape[0], a[2].shape[1] - 1
        skip = (kw.get("zero_tail", False) and a[0].dtype in (torch.uint8, torch.int16) and (ss in (64, 128) or ss % 256 == 0)
                and not kw.get("want_raw") and not kw.get("want_sample_feat") and not kw.get("feat_max"))
        rec["events"].append((e0, e1, rr * (ss // 2 + 1) if skip else rr * ss))
        return out

    def bracket(fn):
        """W warm-up steps happened before; EXACTLY the K steps of fn() between barrier + synchronize brackets; returns the
        max-over-ranks wall time."""
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        rec["on"] = kprobe.on = True
        t0 = time.perf_counter()
        with steady_gc():  # (what the product's own loops do: no full cyclic collection of the process's resident objects inside a timed region)
            fn()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], device="cpu" if share else dev, dtype=torch.float64)
        rec["on"] = kprobe.on = False
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def region_a(renderer, Ksteps=Ksteps, Wsteps=Wsteps, lean=False, hw=None):
        """Render-only region.  Returns (elapsed, kernel events of the timed steps)."""
        h_, w_ = hw or (H, W)
        kmat = synth.intrinsics(h_, w_)
        n_rec = (Ksteps + Wsteps) * Q
        records = torch.zeros(n_rec, 20, device=dev)

        def step(i):
            q0 = (i * world + rank) * Q  # batches of Q consecutive queries, dealt round-robin over ranks
            c2ws = torch.stack([poses[(q0 + j) % 64] for j in range(Q)])
            out = renderer.render_novel_views((h_, w_), kmat, c2ws, unnorm, dev, lean=lean, want_im_pred=True)
            r_ = records[i * Q:(i + 1) * Q]
            r_[:, 0] = torch.arange(q0, q0 + Q, device=dev)
            r_[:, 1:17] = c2ws.reshape(Q, 16).to(dev, non_blocking=True)
            r_[:, 17] = out["pt_feat"][:, 0, 0]

        for i in range(Wsteps):
            step(i)
        rec["events"] = []

        def timed():
            for i in range(Ksteps):
                step(Wsteps + i)
            if use_dist:
                src = records.cpu() if share else records  # gloo gathers host tensors only
                gathered = [torch.empty_like(src) for _ in range(world)]
                dist.all_gather(gathered, src)

        el = bracket(timed)
        ev_, rec["events"] = rec["events"], []
        return el, ev_

    def region_b(renderer, hw=None, Ksteps=Ksteps, Wsteps=Wsteps, style=None, info=None):
        """The evaluator's localisation loop (lean render + c2f matcher) over K batches of Q queries per rank."""
        from nerfmatch_amd.bench_match import CodedRenderer, build_evaluator

        h_, w_ = hw or (H, W)
        ev, make_batch = build_evaluator(dev, h_, w_, queries=Q, style=style)
        if style == "peaked":
            renderer = CodedRenderer(renderer, ev.peaked_code)
        kw = dict(renderer=renderer, solver="none", query2query=True, mutual=True)
        ev.eval_data_loader(data_loader=Batches(Wsteps * world, 0, Q, poses, unnorm, make_batch), **kw)
        timed_loader = Batches(Ksteps * world, Wsteps * world, Q, poses, unnorm, make_batch)
        res = {}
        el = bracket(lambda: res.update(out=ev.eval_data_loader(data_loader=timed_loader, **kw)))
        assert len(res["out"]["query_idx"]) == Ksteps * world * Q  # every rank holds the records of ALL queries
        if info is not None:
            info.update(matches_per_query=float(res["out"]["num_matches"].mean()), matches_min=int(res["out"]["num_matches"].min()),
                        matches_max=int(res["out"]["num_matches"].max()))
        return el

    ops.nerf_fwd = timed_fwd
    rmod.ops.nerf_fwd = timed_fwd
    extra = not args.no_extra_legs
    bf = args.precision in ("bf16x3", "fp16x3")  # the split kernels (three 16-bit MFMAs per product)
    mprec = "fp32" if args.precision == "fp32" else "bf16x3"  # arithmetic of the matcher's contractions (nerfmatch_amd.set_precision)

    # ---- region A (metric i, `value`): EVERY sample of both passes goes through the MLP, like the reference
    ren.skip_zero_tail = False
    elapsed, main_events = region_a(ren)
    # the other arithmetic path, same region definition (extra fields, never `value`)
    other = "fp32" if bf else "fp16x3"
    elapsed_other = other_events = None
    bf16leg = None
    skipleg = None
    if extra:
        ren.precision = other
        elapsed_other, other_events = region_a(ren)
        ren.precision = args.precision
        if args.precision == "fp16x3":  # the bf16-split kernel of rounds 1-2, same region
            ren.precision = "bf16x3"
            bf16leg = region_a(ren)
            ren.precision = args.precision
        # the same region with the fine pass skipping the zero-width intervals (identical outputs; rounds 1-2 reported this as `value`)
        if bf:
            ren.skip_zero_tail = True
            skipleg = region_a(ren)
    ren.skip_zero_tail = True  # the renderer's default from here on (the evaluator's lean render uses it: identical outputs)
    # ---- extra leg: the Cambridge NeRF (appearance embedding + white background), same region A
    cam = None
    if extra and args.variant != "cambridge":
        ren_c, _ = make_renderer("cambridge")
        ren_c.skip_zero_tail = False
        el_c, ev_c = region_a(ren_c)
        cam = (el_c, ev_c)
        del ren_c
    # ---- extra leg: the headline region on TRAINED-LIKE weights (synth.SURFACE_STYLE: hidden activations ~20, densities in the thousands,
    # opacity saturating within 2-4 samples) -- the regime every parity argument is about; operand DATA matter at the power limit
    trained = None
    if extra and args.precision == "fp16x3":
        ren_t, _ = make_renderer(args.variant, style="surface")
        ren_t.skip_zero_tail = False
        el_t, ev_t = region_a(ren_t)
        sat_c, _ = ren_t.nerf_coarse.packed(dev, "fp16x3").nm_guard.read()
        sat_f, _ = ren_t.nerf_fine.packed(dev, "fp16x3").nm_guard.read()
        trained = (el_t, ev_t, bool(sat_c or sat_f), ren_t.calibrate(dev))
        del ren_t
    # ---- extra leg: exactly what the reference's render_novel_view RETURNS (im_pred, pt3d, pt_feat: renderer.py:315-333) -- the
    # fine pass with all its heads, the coarse pass reduced to the compositing weights that place the fine samples
    contract = None
    if extra and bf:
        el_c, ev_c = region_a(ren, lean=True)
        contract = (el_c, ev_c)
    # ---- extra leg: the same NeRF at 256 + 256 samples per ray (BASELINE config 5's ray length), a quarter of the steps
    cam256 = None
    if extra and S != 256:
        k256 = max(2, Ksteps // 4)
        ren_c, _ = make_renderer("cambridge", 256)
        ren_c.skip_zero_tail = False
        el_c, ev_c = region_a(ren_c, k256, 1)
        cam256 = (el_c, ev_c, k256)
        del ren_c
    # ---- extra leg: the reference's shipped geometry (480x480 -> 3600 rays, 128 + 128 samples), half the steps
    refgeo = None
    REF_HW, REF_S = (480, 480), 128
    if extra and ((H, W) != REF_HW or S != REF_S):
        kref = max(2, Ksteps // 2)
        ren_r, _ = make_renderer(args.variant, REF_S)
        ren_r.skip_zero_tail = False
        el_r, ev_r = region_a(ren_r, kref, 1, hw=REF_HW)
        refgeo = dict(a=(el_r, ev_r, kref))
        ren_r.skip_zero_tail = True
    # ---- extra leg (BASELINE config 3, "bf16" throughput configuration: 16-bit operands, ONE product per block): the whole
    # render on the single-product fp16 kernel, with its error against the fp32-MFMA kernel on identical inputs in the line
    single = None
    if extra and bf:
        ren.precision, ren.skip_zero_tail = "fp16x1", False
        el_s, ev_s = region_a(ren)
        c2ws = torch.stack(poses[:2])
        Rq = 2 * R
        tr, jt = torch.rand(Rq, S + 1, device=dev), torch.rand(Rq, S + 1, device=dev) * (1.0 / (S + 1) - 1.2e-7)
        o16 = ren.render_novel_views((H, W), synth.intrinsics(H, W), c2ws, unnorm, dev, lean=False, t_rand=tr, jitter=jt)
        ren.precision = "fp32"
        o32 = ren.render_novel_views((H, W), synth.intrinsics(H, W), c2ws, unnorm, dev, lean=False, t_rand=tr, jitter=jt)
        ren.precision, ren.skip_zero_tail = args.precision, True
        err = {k: float((o16[k] - o32[k]).abs().max() / max(1.0, float(o32[k].abs().max()))) for k in ("pt_feat", "pt3d", "im_pred")}
        single = (el_s, ev_s, err)
    ops.nerf_fwd = raw_fwd
    rmod.ops.nerf_fwd = raw_fwd

    # ---- region B (metric ii): the evaluator's localisation loop, both passes of its lean render on `--precision`
    elapsed_loc = elapsed_loc_fp16 = attn_stats = None
    if not args.no_match:
        nerfmatch_amd.set_precision(mprec)  # the matcher's contractions follow the same arithmetic choice
        assert ren.coarse_precision == "same"
        kprobe.events.clear()
        elapsed_loc = region_b(ren)
        attn_stats = kprobe.take("attn32_v3_kernel", min_flop=4.0 * R * R * 256)  # the 4800^2 layers (the fine stage's 25-token windows run another kernel)
        kprobe.events.clear()
        if extra and bf:  # opt-in: coarse pass of the lean render on one fp16 product
            ren.coarse_precision = "fp16x1"
            elapsed_loc_fp16 = region_b(ren, Wsteps=max(1, Wsteps // 2))
            ren.coarse_precision = "same"
        if refgeo is not None:
            refgeo["b"] = (region_b(ren_r, hw=REF_HW, Ksteps=kref, Wsteps=1), kref)
        # ---- extra leg (round 6): region B in the regime a TRAINED matcher produces -- thousands of mutual matches per query instead of the
        # ~160 of random weights: aligned weights at temperature 30, planted correspondences (bench_match.build_evaluator(style="peaked"))
        peaked = None
        if extra:
            kp_ = max(2, Ksteps // 2)
            pinfo = {}
            el_p = region_b(ren, Ksteps=kp_, Wsteps=2, style="peaked", info=pinfo)
            peaked = dict(elapsed=el_p, steps=kp_, **pinfo)
        # ---- extra leg (BASELINE config 5): Cambridge NeRF at 256 + 256 samples per ray + the matcher's attention on fp8 MFMA
        fp8leg = None
        if extra and bf:
            k5 = max(2, Ksteps // 4)
            ren5, _ = make_renderer("cambridge", 256)
            el_ref = region_b(ren5, Ksteps=k5, Wsteps=1)
            ops.ATTENTION_PRECISION = "fp8"
            el_f8 = region_b(ren5, Ksteps=k5, Wsteps=1)
            # error of the fp8 kernel against the fp32-MFMA attention kernel on one 4800 x 4800, 8-head problem (N(0,1) inputs)
            g = torch.Generator(device="cpu").manual_seed(4)
            q8, k8, v8 = (torch.randn(1, R, 256, generator=g).to(dev) for _ in range(3))
            a8 = ops.attention(q8, k8, v8, 8, 32**-0.5)
            ops.ATTENTION_PRECISION = "fp32"
            a32 = ops.attention(q8, k8, v8, 8, 32**-0.5)
            d8 = (a8 - a32).abs()
            fp8leg = (el_ref, el_f8, k5, float(d8.max()), float(d8.pow(2).mean().sqrt()), float(a32.pow(2).mean().sqrt()))
            del ren5
        nerfmatch_amd.set_precision("fp32")

    # ---- extra leg: the reference's own operating point -- ONE query per step (its loop is batch 1: nerfmatch_evaluator.py:631-724; it
    # times match_time / localize_time per query, :150-230, :502-629).  Wall time of eval_batch (lean render_novel_view + matcher forward)
    # with a synchronize on both sides of every step, median of 30, beside the GPU time and the count of the step's native calls
    latency_q1 = None
    if extra and not args.no_match and rank == 0:
        from nerfmatch_amd import latency

        nerfmatch_amd.set_precision(mprec)
        latency_q1 = {}
        for kind_ in ("c2f", "coarse"):
            m_ = latency.measure(dev, ren, H, W, kind=kind_, n=30, queries=1, warmup=5)
            latency_q1[kind_] = {k: v for k, v in m_.items() if k not in ("per_call", "series")}
            print(f"[bench] one-query steps ({kind_}), wall ms in order: " + " ".join(f"{t:.2f}" for t in m_["series"]), file=sys.stderr)
            latency_q1[kind_]["top_calls_ms"] = {k: round(v[1], 4) for k, v in sorted(m_["per_call"].items(), key=lambda kv: -kv[1][1])[:6]}
        if peaked is not None:  # the one-query twin of the peaked leg
            m_ = latency.measure(dev, ren, H, W, kind="c2f", n=20, queries=1, warmup=5, style="peaked")
            peaked["q1"] = {k: v for k, v in m_.items() if k not in ("per_call", "series")}
            peaked["q1"]["fine_stage_ms"] = m_["per_call"].get("nm_fine_stage", (0, 0.0))[1]
            peaked["q1"]["top_calls_ms"] = {k: round(v[1], 4) for k, v in sorted(m_["per_call"].items(), key=lambda kv: -kv[1][1])[:6]}
        if not use_dist:  # the evaluator's LOOP at batch 1 (eval_data_loader: step i+1's host work overlaps step i's kernels; one sync per batch)
            from nerfmatch_amd.bench_match import build_evaluator

            ev1, mk1 = build_evaluator(dev, H, W, queries=1)
            kw1 = dict(renderer=ren, solver="none", query2query=True, mutual=True)
            # round 6: the loop as shipped (query i+1's render on five whole XCDs beside query i's matcher on the other three) and, beside it,
            # the one-stream loop of round 5 -- same kernels, same bits (tests/test_evaluator_gpu.py), best of three runs of 40 queries each
            for key_, on_ in (("loop_one_stream_ms_per_query", False), ("loop_ms_per_query", True)):
                ev1.overlap_render = on_
                ev1.eval_data_loader(data_loader=Batches(5, 0, 1, poses, unnorm, mk1), **kw1)
                best_ = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    ev1.eval_data_loader(data_loader=Batches(40, 5, 1, poses, unnorm, mk1), **kw1)
                    torch.cuda.synchronize()
                    el_ = (time.perf_counter() - t0_) / 40 * 1e3
                    best_ = el_ if best_ is None else min(best_, el_)
                latency_q1[key_] = best_
            latency_q1["loop_partitions"] = {"render": list(ev1.render_part) if isinstance(ev1.render_part, tuple) else ev1.render_part,
                                             "matcher": list(ev1.match_part) if isinstance(ev1.match_part, tuple) else ev1.match_part}
            if peaked is not None:  # ... and the loop at batch 1 in the peaked regime
                from nerfmatch_amd.bench_match import CodedRenderer

                evp, mkp = build_evaluator(dev, H, W, queries=1, style="peaked")
                kwp = dict(renderer=CodedRenderer(ren, evp.peaked_code), solver="none", query2query=True, mutual=True)
                evp.eval_data_loader(data_loader=Batches(5, 0, 1, poses, unnorm, mkp), **kwp)
                best_ = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                    evp.eval_data_loader(data_loader=Batches(40, 5, 1, poses, unnorm, mkp), **kwp)
                    torch.cuda.synchronize()
                    el_ = (time.perf_counter() - t0_) / 40 * 1e3
                    best_ = el_ if best_ is None else min(best_, el_)
                peaked["q1"]["loop_ms_per_query"] = best_
                peaked["q1"]["loop_spec_batches"], peaked["q1"]["loop_spec_reruns"] = getattr(evp.model, "spec_batches", 0), getattr(evp.model, "spec_reruns", 0)
        nerfmatch_amd.set_precision("fp32")
    if use_dist:
        dist.barrier()


    # ---- extra legs: the SURVEY 8f rows, driver-timed (never `value`): a few timed steps each, the same barrier / synchronize brackets
    next_rows = {}
    if extra and not args.no_match:
        import tempfile
        from argparse import Namespace
        import numpy as np
        from nerfmatch_amd import inerf
        from nerfmatch_amd.matcher import NeRFMatcherMS
        from nerfmatch_amd.modules import PrecomputedBackbone
        from nerfmatch_amd.nerf_evaluator import NerfEvaluator

        nerfmatch_amd.set_precision(mprec)
        kmat = synth.intrinsics(H, W)
        g = torch.Generator().manual_seed(9)
        # (f1) iNeRF refinement: Adam steps on the pose through the fine network, 128 + 128 samples as the reference hard-codes
        # (nerfmatch_evaluator.py:354,360), one 640x480 query = 4800 rays
        ren_i, _ = make_renderer(args.variant, 128)
        img_i = torch.rand(H, W, 3, generator=g).to(dev)
        pose_i = torch.as_tensor(synth.camera_pose(1), dtype=torch.float32).to(dev)
        inerf.refine(ren_i, kmat, H, W, img_i, pose_i, num_optim=2)
        n_i = 8
        # (three refinements of 8 steps, the median: one in ten of these short regions catches a host hiccup worth a millisecond per step)
        els_i = sorted(bracket(lambda: inerf.refine(ren_i, kmat, H, W, img_i, pose_i, num_optim=n_i)) for _ in range(3))
        el_i = els_i[1]
        next_rows["inerf_step_ms"] = {"value": el_i / n_i * 1e3, "unit": "ms/step", "steps_timed": n_i, "repeats_ms_per_step": [round(e_ / n_i * 1e3, 3) for e_ in els_i],
                                      "workload": f"inerf.refine (nerfmatch_evaluator.py:288-500): {R} rays x (128+128) samples, photometric loss, Adam on the 4x4 pose; "
                                                  f"coarse pass {args.precision}, fine pass forward + backward (DESIGN 3.7)"}
        # the same refinement WITH the matching term (use_match_loss, nerfmatch_evaluator.py:429-448): every step also runs the c2f matcher's
        # training-mode forward and its backward to the rendered features / points; that share is timed by HIP events around it
        if args.variant == "7scenes":
            from nerfmatch_amd.bench_match import build_evaluator as _be
            ev_i, _ = _be(dev, H, W, queries=1)
            match_i = dict(model=ev_i.model, image=torch.zeros(1, 3, H, W, device=dev), im_mask=torch.ones(1, R, dtype=torch.bool, device=dev),
                           pt_mask=torch.ones(1, R, dtype=torch.bool, device=dev), unnorm=unnorm.to(dev))
            raw_mt, spent_mt = inerf._match_term, []

            def timed_mt(*a_, **k_):
                e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0_.record()
                o_ = raw_mt(*a_, **k_)
                e1_.record()
                spent_mt.append((e0_, e1_))
                return o_

            inerf._match_term = timed_mt
            try:
                inerf.refine(ren_i, kmat, H, W, img_i, pose_i, num_optim=3, match=match_i)
                spent_mt.clear()
                n_im = 8
                el_im = sorted(bracket(lambda: inerf.refine(ren_i, kmat, H, W, img_i, pose_i, num_optim=n_im, match=match_i)) for _ in range(3))[1]  # (median of three, as above)
            finally:
                inerf._match_term = raw_mt
            mt_ms = sum(a_.elapsed_time(b_) for a_, b_ in spent_mt) / max(1, len(spent_mt))
            # one more call of the matcher's share with HIP events around every native call (outside the timed region): count and largest spans
            from nerfmatch_amd import latency as _lat
            pf_i = torch.relu(torch.randn(R, 256, generator=g)).to(dev)
            p3_i = (torch.randn(R, 3, generator=g) * 0.25).to(dev)
            with _lat.timed_lib() as tl_:
                tl_.spans = []
                inerf._match_term(match_i, pf_i, p3_i)
                torch.cuda.synchronize()
                per_ = {}
                for nm_, e0_, e1_ in tl_.spans:
                    c_ = per_.setdefault(nm_, [0, 0.0])
                    c_[0] += 1
                    c_[1] += e0_.elapsed_time(e1_)
                mt_calls = len(tl_.spans)
            mt_top = {k_: [v_[0], round(v_[1], 3)] for k_, v_ in sorted(per_.items(), key=lambda kv: -kv[1][1])[:8]}
            next_rows["inerf_match_step_ms"] = {
                "value": el_im / n_im * 1e3, "unit": "ms/step", "steps_timed": n_im, "matcher_fwd_bwd_ms": mt_ms, "nerf_side_ms": el_im / n_im * 1e3 - mt_ms,
                "matcher_native_calls": mt_calls, "matcher_top_calls": mt_top,
                "workload": f"inerf.refine with use_match_loss: the step above + NeRFMatcherMS.match_loss on {R} x {R} tokens (training-mode forward, focal loss against "
                            f"the identity, backward to pt_feat / pt3d; parameters frozen) -- `matcher_fwd_bwd_ms` of every step is the matcher itself, which no "
                            f"NeRF-side kernel can shorten; the NeRF side runs the fused kernel pair (the forward kernel writes the tapped layer, the backward kernel takes "
                            f"the term's gradient in at that layer: nm_nerf_points_*_tap_bf16x3), DESIGN 3.7"}
            del ev_i, match_i
        del ren_i
        # (f3) forward_multi_pair: one query against k = 3 reference frames' point sets (image side evaluated once), 4 queries per call
        Bq, kk = 4, 3
        cf = torch.randn(Bq, 256, H // DS, W // DS, generator=g).to(dev)
        ff = torch.randn(Bq, 128, H // 2, W // 2, generator=g).to(dev)
        mp = NeRFMatcherMS(synth.matcher_config("c2f"))
        mp.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
        mp.backbone = PrecomputedBackbone((cf, ff), [256, 128])
        mp.to(dev).eval()
        mp.keep_conf = False
        ptf = torch.relu(torch.randn(Bq, kk, R, 256, generator=g)).to(dev)
        p3 = (torch.randn(Bq, kk, R, 3, generator=g) * 2).to(dev)
        ys, xs = torch.meshgrid(torch.arange(H // DS), torch.arange(W // DS), indexing="ij")
        p2 = (torch.stack([xs, ys], -1).reshape(1, -1, 2).float() * 8 + 4).repeat(Bq, 1, 1).to(dev)
        mkd = lambda: dict(image=torch.zeros(Bq, 3, 8, 8, device=dev), im_mask=torch.ones(Bq, R, dtype=torch.bool, device=dev), pt3d=p3, pt_feat=ptf,
                           pt_mask=torch.ones(Bq, kk, R, dtype=torch.bool, device=dev), pt2d=p2)
        mp.forward(mkd(), mutual=True)
        n_m = 4
        el_m3 = bracket(lambda: [mp.forward(mkd(), mutual=True) for _ in range(n_m)])
        next_rows["multi_pair_k3_ms"] = {"value": el_m3 / (n_m * Bq) * 1e3, "unit": "ms/query", "queries_timed": n_m * Bq,
                                         "workload": f"NeRFMatcherMS.forward on multi-pair batches (c2f_trainer.py:371-427): {Bq} queries x k = {kk} point sets of {R} points, "
                                                     f"{R} image tokens, mutual NN + fine stage, {mprec} contractions; image side hoisted out of the k loop"}
        del mp, ptf, p3
        # (f4) one training step of the matcher head: forward + backward + AdamW, B = 2 pairs of 3600 + 3600 tokens (480x480)
        Ht, Wt, Bt = 480, 480, 2
        ht, wt = Ht // DS, Wt // DS
        Mt = ht * wt
        cft, fft = torch.randn(Bt, 256, ht, wt, generator=g).to(dev), torch.randn(Bt, 128, Ht // 2, Wt // 2, generator=g).to(dev)
        ptft, p3t = torch.relu(torch.randn(Bt, Mt, 256, generator=g)).to(dev), (torch.randn(Bt, Mt, 3, generator=g) * 2).to(dev)
        cgt = torch.zeros(Bt, Mt, Mt, dtype=torch.bool)
        for b_ in range(Bt):
            cgt[b_, torch.arange(Mt // 2), torch.randperm(Mt, generator=g)[: Mt // 2]] = True
        cgt = cgt.to(dev)
        ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wt), indexing="ij")
        p2t = (torch.stack([xs, ys], -1).reshape(1, -1, 2).float() * 8 + 4).repeat(Bt, 1, 1).to(dev)
        p2p = (torch.rand(Bt, Mt, 2, generator=g) * torch.tensor([Wt, Ht])).to(dev)
        mt = NeRFMatcherMS(synth.matcher_config("c2f"))
        mt.load_state_dict(synth.matcher_state_dict("c2f", seed=0), strict=False)
        mt = mt.to(dev)
        mt.backbone = PrecomputedBackbone((cft, fft), [256, 128])
        np.random.seed(0)
        with torch.enable_grad():
            opt = torch.optim.AdamW(mt.parameters(), lr=1e-4)

            def train_step():
                d_ = dict(image=torch.zeros(Bt, 3, 8, 8, device=dev), im_mask=torch.ones(Bt, Mt, dtype=torch.bool, device=dev),
                          pt_mask=torch.ones(Bt, Mt, dtype=torch.bool, device=dev), pt3d=p3t, pt2d=p2t, conf_gt=cgt, pt2d_proj=p2p, pt_feat=ptft)
                m_ = mt.forward_with_metrics(d_, training=True)
                opt.zero_grad()
                m_["loss"].backward()
                opt.step()

            for _ in range(2):
                train_step()
            n_t = 5
            el_t = sorted(bracket(lambda: [train_step() for _ in range(n_t)]) for _ in range(3))[1]  # (median of three regions of n_t steps)
        next_rows["train_step_ms"] = {"value": el_t / n_t * 1e3, "unit": "ms/step", "steps_timed": n_t,
                                      "workload": f"NeRFMatcherMS.forward_with_metrics(training) + backward + AdamW (c2f_trainer.py:490-551), B = {Bt} pairs of {Mt} + {Mt} tokens "
                                                  f"({Wt}x{Ht}), GT-padded coarse + fine loss, {mprec} contractions (DESIGN 3.8)"}
        del mt, opt, cgt, cft, fft
        # (f2) scene-feature cache: frames rendered and written in the reference's per-frame .npy format (nerf_evaluator.py:308-402)
        cfg_c = synth.nerf_config(args.variant, num_pts=S, img_wh=(W, H))
        cfg_c.exp, cfg_c.split, cfg_c.downsample = Namespace(seed=0), "train", DS
        nfr = 32
        frames = []
        for f_ in range(nfr):
            rays_f, _ = ops.raygen(kmat, synth.camera_pose(f_), H, W, dev)
            frames.append(dict(img_wh=torch.tensor([[W // DS, H // DS]]), rays=rays_f[None], rgbs=torch.zeros(1, R, 3), img_idx=[f"seq1_frame{f_:05d}"],
                               unnorm_scene=unnorm[None], **({"ts": torch.ones(1, R, dtype=torch.long)} if args.variant == "cambridge" else {})))
        evc = NerfEvaluator(cfg_c, vocab_num=8, stop_layer=3, data_loader=frames)
        evc.model.load_state_dict(sd, strict=True)
        evc.model.precision = args.precision
        with tempfile.TemporaryDirectory() as td:
            evc.cache_scene_pts(cache_dir=Path(td) / "warm", frames_per_launch=4)
            el_c2 = bracket(lambda: evc.cache_scene_pts(cache_dir=Path(td) / "timed", frames_per_launch=4))
        next_rows["cache_frames_per_s"] = {"value": world * nfr / el_c2, "unit": "frames/s", "frames_timed": nfr,
                                           "workload": f"NerfEvaluator.cache_scene_pts: {nfr} frames of {R} rays x ({S}+{S}) samples per rank, lean render on {args.precision} "
                                                       "(4 frames per launch), read-back and one pickled-dict .npy per frame written to a temporary directory (file I/O inside the region)"}
        del evc, frames
        nerfmatch_amd.set_precision("fp32")

    # ---- extra leg: NeRFMatch-Mini (BASELINE config 2): coarse-only model = 4800 x 4800 dual-softmax + mutual NN
    mini = None
    if extra and not args.no_match:
        from nerfmatch_amd.matcher import NeRFMatcherCoarse
        from nerfmatch_amd.modules import PrecomputedBackbone

        mm = NeRFMatcherCoarse(synth.matcher_config("coarse"))
        mm.load_state_dict(synth.matcher_state_dict("coarse"), strict=False)
        im, pt = synth.separated_features(R, R, 256, seed=2)
        cf = im.T.reshape(1, 256, H // DS, W // DS).expand(Q, -1, -1, -1).contiguous().to(dev)
        mm.backbone = PrecomputedBackbone(cf, 256)
        mm.to(dev).eval()
        mm.keep_conf = False  # as the evaluator runs it (NeRFMatchEvaluator.keep_conf_matrix = False): match lists only
        nerfmatch_amd.set_precision(mprec)
        data = lambda: dict(image=torch.zeros(Q, 3, 8, 8, device=dev), im_mask=torch.ones(Q, R, dtype=torch.bool, device=dev),
                            pt3d=torch.zeros(Q, R, 3, device=dev), pt_feat=pt[None].expand(Q, -1, -1).contiguous().to(dev),
                            pt_mask=torch.ones(Q, R, dtype=torch.bool, device=dev), pt2d=None)
        for _ in range(max(1, Wsteps)):
            mm.forward(data(), mutual=True)
        d_ = data()
        nmatch = {}

        def mini_steps():
            for _ in range(Ksteps):
                nmatch["n"] = int(mm.forward(d_, mutual=True)["match_ids"][0].shape[0])

        kprobe.events.clear()
        el_m = bracket(mini_steps)
        fused_stats = kprobe.take("nm_dual_softmax_match_fused")
        mm.keep_conf = True  # the reference's contract: conf_matrix (92 MB per 4800^2 pair) is written into the batch dict
        mm.forward(data(), mutual=True)
        el_mc = bracket(mini_steps)
        nerfmatch_amd.set_precision("fp32")
        per_pair = el_m / (Ksteps * Q)
        mini = {"metric": f"image/point-set pairs per second, coarse-only matcher (NeRFMatch-Mini): {R} x {R} dual-softmax + mutual NN",
                "value": world * Ksteps * Q / el_m, "unit": "pairs/s", "ms_per_pair": per_pair * 1e3, "matches_per_step": nmatch.get("n"),
                "ms_per_pair_with_conf_matrix": el_mc / (Ksteps * Q) * 1e3,
                "mode": "match lists only (keep_conf = False, what the evaluator runs): similarity / confidence stay in registers, one launch sequence per "
                        "batch (csrc/match_fused.hip); `ms_per_pair_with_conf_matrix` = the same call returning conf_matrix like the reference's forward",
                }
        if fused_stats is not None:
            # csrc/match_fused.hip never writes the similarity matrix: it computes the 128 x 128 tiles TWICE on the 16-bit matrix cores (three
            # products per fp32 product) -- the path is MFMA-bound, and every figure below follows from what the kernels do
            n_call, ms_call, flop_sum, sec_sum = fused_stats
            alg = flop_sum / sec_sum / 1e12
            mini["roofline"] = {
                "bound": "mfma", "kernel": "match_tile_kernel<1> + match_tile_kernel<2> (+ norm_pack, merge, select, tie, compact: one nm_dual_softmax_match_fused call per batch)",
                "achieved": alg, "peak": PEAK_TFLOPS["bf16x3"], "unit": "TFLOP/s", "frac": alg / PEAK_TFLOPS["bf16x3"],
                "achieved_note": "algorithmic FLOP 2*M*N*C per pair (one similarity matrix, SURVEY 8d) / mean duration of the call's launches (HIP events on the launch stream)",
                "executed_mfma_tflops": 6.0 * alg, "frac_executed": 6.0 * alg / PEAK_TFLOPS["bf16x3"],
                "executed_note": "issued 16-bit MFMA FLOP: 2 passes over the tiles x 3 products per fp32 product",
                "flop_per_call": flop_sum / n_call, "avg_call_ms": ms_call, "calls_timed": n_call, "pairs_per_call": Q,
                "traffic": pmc_traffic(["r5_pmc_match_tile1.json", "r5_pmc_match_tile2.json"], Q == 16 and (H, W) == (480, 640)),
                "traffic_algorithmic": 2.0 * 2 * R * 256 * 4 * Q,
                "traffic_note": "L2<->fabric bytes of the two tile passes of ONE 16-pair call (rocprofv3 PMC passes of scripts/pmc_mini.py on these kernels -- unchanged "
                                "since round 5 --, FETCH_SIZE x2-corrected + WRITE_SIZE, summed over the passes; null at other batch sizes); traffic_algorithmic = both "
                                "operand sets (2 x 4800 x 256 fp32 per pair) read once per pass",
                "hbm_note": f"for scale: SURVEY 8d's conf-materialised bytes 8*M*N per pair / this time = {8.0 * R * R * Q / (ms_call * 1e-3) / 1e9:.0f} GB/s "
                            "-- bytes this path does not move (operands: 2 x 4.9 MB per pair and pass); profiles/r5_pmc_match_tile*.json"}

    # ---- the JSON line
    def kernel_stats(events, flop_per_sample):
        ms = [a.elapsed_time(b) for a, b, _ in events]
        n = [c for _, _, c in events]
        avg_s = sum(ms) / len(ms) * 1e-3
        evald = sum(n) / len(n)
        return avg_s, evald, evald * flop_per_sample / avg_s / 1e12, len(ms)

    traffic = None
    traffic_scaled = True
    # round 6: the counters collected AT THE BENCH'S LAUNCH SIZE (16 queries per launch, scripts/pmc_render_q16.py) -- not a one-query figure times 16
    q16 = ROOT / "profiles" / "r6_pmc_nerf_fwd_fp16x3_q16.json"
    # (the fp16x3 and bf16x3 kernels are one template with identical memory behaviour: a bf16x3 PMC pass stands in until an fp16x3 one exists)
    sfx = {"fp32": [".json"], "bf16x3": ["_bf16x3.json"], "fp16x3": ["_fp16x3.json", "_bf16x3.json"]}[args.precision]
    for pmc in [ROOT / "profiles" / (name + x) for x in sfx for name in ("r5_pmc_nerf_fwd", "r4_pmc_nerf_fwd", "r3_pmc_nerf_fwd", "r2_pmc_nerf_fwd", "r1_pmc_nerf_fwd")]:
        if pmc.exists() and S == 64 and args.variant == "7scenes":
            traffic = json.load(open(pmc))["derived"]["traffic_bytes"] * Q * R / 4800  # measured per 4800-ray launch; scales with the rays
            traffic_src = pmc.name
            break
    if q16.exists() and args.precision == "fp16x3" and S == 64 and args.variant == "7scenes" and Q == 16 and R == 4800:
        traffic, traffic_src, traffic_scaled = json.load(open(q16))["derived"]["traffic_bytes"], q16.name, False
    if rank == 0:
        fps = FLOP_PER_SAMPLE_PASS[args.variant]
        total_units = world * Ksteps * Q * R * 2 * S
        avg_s, evald, achieved, nlaunch = kernel_stats(main_events, fps)
        peak = PEAK_TFLOPS[args.precision]
        line = {
            "metric": "rays*samples/sec",
            "value": total_units / elapsed,
            "unit": "rays*samples/s",
            "n_gpus": world,
            "steps": Ksteps,
            "warmup": Wsteps,
            "ms_per_step": elapsed / Ksteps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": (f"{args.precision} ({'fp16' if args.precision == 'fp16x3' else 'bf16'} MFMA on hi/lo-split fp32 operands, three products per fp32 product, "
                      "fp32 accumulate; fp32 everywhere else)") if bf else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{Q} {args.variant}-style queries per rank and step (one batch): render_novel_views {W}x{H} ds8 -> {Q}x{R} rays x ({S}+{S}) samples "
                            f"(coarse+fine 8x256 NeRF, stop_layer 3, ret_pfeat, all reference outputs, EVERY sample evaluated, {args.precision} kernel) "
                            f"[timed region of `value`]; "
                            f"query_images_per_sec = a second timed region of the same K steps through NeRFMatchEvaluator.eval_data_loader (solver none, query2query): "
                            f"render of pt3d / pt_feat only (both passes on the {args.precision} kernel; the fine pass skips the zero-width intervals the reference's "
                            f"resampler leaves -- weight exactly 0, outputs identical, checked on the device per launch) "
                            f"+ the c2f matcher ({R}x{R} tokens, mutual NN, fine stage; image backbone and PnP excluded), "
                            f"batches pipelined across the matcher's one synchronisation point",
                "rays": R, "samples_coarse": S, "samples_fine": S, "queries_per_step_per_gpu": Q, "variant": args.variant, "image_hw": [H, W],
                "sharding": "query batches round-robin over ranks; one all_gather of pose-candidate records at shard end",
                "world_size": world, "collectives": ("gloo, all ranks on ONE GPU (NM_BENCH_SHARE_GPU dry run: not a measurement)" if share else "RCCL (torch.distributed nccl)") if use_dist else "none (single process)",
            },
            "query_images_per_sec": (world * Ksteps * Q / elapsed_loc) if elapsed_loc else None,
            "localize_ms_per_query": (elapsed_loc / (Ksteps * Q) * 1e3) if elapsed_loc else None,
            "roofline": {
                "bound": "mfma", "kernel": KERNEL[args.precision], "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "achieved_note": "fp32-equivalent FLOP of the samples a launch evaluates (all of them in this region; FLOP per sample and pass from SURVEY 8d) / mean launch duration",
                "evaluated_samples_per_launch": evald,
                "traffic_note": (f"L2<->fabric bytes per launch from rocprofv3 PMC passes (profiles/{traffic_src}), FETCH_SIZE x2-corrected + WRITE_SIZE, "
                                 + ("scaled by the ray count" if traffic_scaled else "collected at this launch size (16 queries x 4800 rays x 64 samples; coarse and fine launches averaged)")
                                 + "; algorithmic bytes are ~55 B per ray in and ~1.1 KB per ray out") if traffic else None,
                "traffic_algorithmic": Q * R * (48 + (S + 1) * 4 + S * 4 + 1024 + 32) + 2621440,
                "flop_per_launch": Q * R * S * fps, "avg_launch_ms": avg_s * 1e3, "launches_timed": nlaunch,
                "launches_note": "HIP events around the launches of the K timed steps only (2 per step: coarse + fine)",
            },
        }
        if bf:
            ex = evald * BF16X3_EXEC_FLOP_PER_SAMPLE_PASS / avg_s / 1e12
            line["roofline"].update(executed_mfma_tflops=ex, frac_executed=ex / peak,
                                    executed_note="16-bit MFMA FLOP actually issued: 3 per fp32 product (w_hi*x_hi + w_hi*x_lo + w_lo*x_hi), padded K")
            sustained = mfma_probe(dev)
            if sustained:
                line["roofline"].update(peak_sustained=sustained, frac_executed_of_sustained=ex / sustained,
                                        peak_sustained_note="measured on THIS box after the timed regions: a bare v_mfma_f32_32x32x16_f16 stream on every SIMD "
                                                            "(nm_probe_mfma_f16, ~50 ms) -- the matrix rate the chip sustains at its power limit; "
                                                            "`peak` is the data-sheet figure at 2.4 GHz and stays the denominator of `frac`")
        if other_events:
            o_s, o_n, o_ach, _ = kernel_stats(other_events, fps)
            line["other_precision"] = {"precision": other, "kernel": KERNEL[other], "value": total_units / elapsed_other, "unit": "rays*samples/s",
                                       "avg_launch_ms": o_s * 1e3, "achieved_tflops": o_ach, "frac_of_peak": o_ach / PEAK_TFLOPS[other],
                                       "peak": PEAK_TFLOPS[other]}
        variants = {}
        if bf16leg is not None:
            b_s, b_n, b_ach, b_l = kernel_stats(bf16leg[1], fps)
            variants["bf16x3"] = {"workload": "region A on the bf16-split kernel of rounds 1-2 (16 mantissa bits: parity-class on smooth fields only, DESIGN 3.1b)",
                                  "value": total_units / bf16leg[0], "unit": "rays*samples/s", "ms_per_step": bf16leg[0] / Ksteps * 1e3,
                                  "roofline": {"bound": "mfma", "kernel": KERNEL["bf16x3"], "achieved": b_ach, "peak": peak, "unit": "TFLOP/s",
                                               "frac": b_ach / peak, "avg_launch_ms": b_s * 1e3, "launches_timed": b_l}}
        if skipleg is not None:
            k_s, k_n, k_ach, k_l = kernel_stats(skipleg[1], fps)
            variants["zero_tail_skip"] = {
                "workload": "region A with the fine pass running the MLP on samples 0..S/2 only: the reference's randomized resampler leaves the other intervals with zero "
                            "width = weight exactly 0 (premise verified on the device per launch), outputs identical; nominal units R*2S / time (rounds 1-2 reported this as `value`)",
                "value": total_units / skipleg[0], "unit": "rays*samples/s", "ms_per_step": skipleg[0] / Ksteps * 1e3,
                "value_evaluated": world * sum(c for _, _, c in skipleg[1]) / skipleg[0],
                "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": k_ach, "peak": peak, "unit": "TFLOP/s", "frac": k_ach / peak,
                             "avg_launch_ms": k_s * 1e3, "evaluated_samples_per_launch": k_n, "launches_timed": k_l}}
        if elapsed_loc_fp16:
            variants["coarse_fp16x1"] = {
                "workload": "region B with the coarse pass of the lean render on ONE fp16 MFMA per product block (NerfRenderer.coarse_precision = 'fp16x1', opt-in; "
                            "narrower arithmetic than the reference: measured on the trained-like fixture in tests/test_nerf_gpu.py::test_surface_fp16x1_coarse_pass_measured)",
                "query_images_per_sec": world * Ksteps * Q / elapsed_loc_fp16, "localize_ms_per_query": elapsed_loc_fp16 / (Ksteps * Q) * 1e3}
        if refgeo is not None:
            el_r, ev_r, kref = refgeo["a"]
            r_s, r_n, r_ach, r_l = kernel_stats(ev_r, fps)
            Rr = (REF_HW[0] // DS) * (REF_HW[1] // DS)
            variants["reference_geometry"] = {
                "workload": f"the reference's shipped geometry: {REF_HW[1]}x{REF_HW[0]} queries -> {Rr} rays / tokens, {REF_S}+{REF_S} samples per ray, {kref} timed steps of {Q} queries; "
                            "region A with every sample evaluated, region B as above",
                "value": world * kref * Q * Rr * 2 * REF_S / el_r, "unit": "rays*samples/s", "ms_per_step": el_r / kref * 1e3,
                "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": r_ach, "peak": peak, "unit": "TFLOP/s", "frac": r_ach / peak,
                             "avg_launch_ms": r_s * 1e3, "launches_timed": r_l},
                "query_images_per_sec": (world * refgeo["b"][1] * Q / refgeo["b"][0]) if "b" in refgeo else None}
        if cam is not None:
            c_s, c_n, c_ach, c_l = kernel_stats(cam[1], FLOP_PER_SAMPLE_PASS["cambridge"])
            variants["cambridge"] = {"workload": f"region A with the Cambridge NeRF (appearance embedding 16, white background), {Q}x{R} rays x ({S}+{S}) samples, every sample evaluated",
                                     "value": total_units / cam[0], "unit": "rays*samples/s", "ms_per_step": cam[0] / Ksteps * 1e3,
                                     "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": c_ach, "peak": peak, "unit": "TFLOP/s",
                                                  "frac": c_ach / peak, "avg_launch_ms": c_s * 1e3, "launches_timed": c_l}}
        if trained is not None:
            t_s, t_n, t_ach, t_l = kernel_stats(trained[1], fps)
            variants["trained_like"] = {
                "workload": "region A (every sample, all outputs) on TRAINED-LIKE weights (synth.SURFACE_STYLE: layer gain 3.2, density head x2600, ~25 % of space "
                            "occupied, fine net sharing the coarse net's density trunk): hidden activations ~20, opacity saturating within 2-4 samples; fp16x3 with "
                            "the operand scales calibrated on the seeded probe bundle; `saturation_flag` = an operand reached +-65504 during the region (must be false)",
                "value": total_units / trained[0], "unit": "rays*samples/s", "ms_per_step": trained[0] / Ksteps * 1e3,
                "saturation_flag": trained[2], "act_log2": trained[3],
                "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": t_ach, "peak": peak, "unit": "TFLOP/s", "frac": t_ach / peak,
                             "avg_launch_ms": t_s * 1e3, "launches_timed": t_l}}
        if contract is not None:
            variants["render_novel_view_outputs_only"] = {
                "workload": f"region A computing only what the reference's render_novel_view returns (im_pred = rgb_fine, pt3d, pt_feat): fine pass with all heads "
                            f"({args.precision}, zero-tail skip), coarse pass = compositing weights only ({args.precision}, density head only); {Q}x{R} rays x ({S}+{S}) samples",
                "value": total_units / contract[0], "unit": "rays*samples/s", "ms_per_step": contract[0] / Ksteps * 1e3,
                "launches_timed": len(contract[1])}
        if cam256 is not None:
            c_s, c_n, c_ach, c_l = kernel_stats(cam256[1], FLOP_PER_SAMPLE_PASS["cambridge"])
            variants["cambridge_s256"] = {"workload": f"region A with the Cambridge NeRF at 256 + 256 samples per ray (BASELINE config 5), {Q}x{R} rays, {cam256[2]} timed steps, every sample evaluated",
                                          "value": world * cam256[2] * Q * R * 512 / cam256[0], "unit": "rays*samples/s", "ms_per_step": cam256[0] / cam256[2] * 1e3,
                                          "roofline": {"bound": "mfma", "kernel": KERNEL[args.precision], "achieved": c_ach, "peak": peak, "unit": "TFLOP/s",
                                                       "frac": c_ach / peak, "avg_launch_ms": c_s * 1e3, "launches_timed": c_l}}
        if single is not None:
            s_s, s_n, s_ach, s_l = kernel_stats(single[1], fps)
            variants["single_product"] = {
                "workload": "BASELINE config 3 ('bf16' throughput configuration: 16-bit operands, ONE fp16 MFMA per product block, fp32 accumulate): region A, "
                            f"both passes and every head on nerf_fwd_fp16x1_kernel, every sample evaluated; {Q}x{R} rays x ({S}+{S}) samples.  NOT parity-class: "
                            "error below = max |difference| to the fp32-MFMA kernel on identical inputs, in units of max(1, max|fp32 value|)",
                "value": total_units / single[0], "unit": "rays*samples/s", "ms_per_step": single[0] / Ksteps * 1e3, "error_vs_fp32_kernel": single[2],
                "roofline": {"bound": "mfma", "kernel": "nerf_fwd_fp16x1_kernel", "achieved": s_ach, "peak": peak, "unit": "TFLOP/s", "frac": s_ach / peak,
                             "avg_launch_ms": s_s * 1e3, "launches_timed": s_l,
                             "note": "one MFMA per product block: here algorithmic FLOP = issued FLOP (up to the K padding)"}}
        if not args.no_match and extra and bf and fp8leg is not None:
            variants["attention_fp8"] = {
                "workload": f"BASELINE config 5: Cambridge NeRF at 256+256 samples per ray + c2f matcher with the attention contractions on ONE e4m3 MFMA per product block "
                            f"(csrc/attention_fp8.hip; everything else as region B), {fp8leg[2]} timed steps of {Q} queries.  NOT parity-class: attention_error = "
                            f"fp8 kernel against the fp32-MFMA attention kernel on one {R}x{R}, 8-head problem with N(0,1) inputs",
                "query_images_per_sec": world * fp8leg[2] * Q / fp8leg[1], "query_images_per_sec_bf16x3_attention": world * fp8leg[2] * Q / fp8leg[0],
                "attention_error": {"max_abs": fp8leg[3], "rms": fp8leg[4], "output_rms": fp8leg[5]}}
        if attn_stats is not None:
            n_call, ms_call, flop_sum, sec_sum = attn_stats
            a_alg = flop_sum / sec_sum / 1e12
            line["roofline_b"] = {
                "region": "B (query_images_per_sec)", "bound": "mfma", "kernel": "attn32_v3_kernel", "achieved": a_alg, "peak": PEAK_TFLOPS["bf16x3"],
                "unit": "TFLOP/s", "frac": a_alg / PEAK_TFLOPS["bf16x3"],
                "achieved_note": "algorithmic FLOP 4*L*S*256 per sequence and layer (Q.K^T and P.V over 8 heads x 32; SURVEY 8d) / mean launch duration, HIP events on the launch "
                                 "stream around every launch of the timed steps (self- and cross-attention layers at 4800 x 4800; the fine stage's 25-token windows run another kernel)",
                "executed_mfma_tflops": 3.0 * a_alg, "frac_executed": 3.0 * a_alg / PEAK_TFLOPS["bf16x3"],
                "executed_note": "issued 16-bit MFMA FLOP: 3 products per fp32 product (operands split into bf16 hi / lo parts)",
                "avg_launch_ms": ms_call, "launches_timed": n_call, "flop_per_launch": flop_sum / n_call,
                "share_of_region_b_time": sec_sum / elapsed_loc,
                "traffic": pmc_traffic(["r5_pmc_attn32_v3.json"]), "traffic_algorithmic": 32 * (3 * 4800 * 256 * 4 + 4800 * 256 * 4),
                "traffic_note": "L2<->fabric bytes of ONE launch of 32 sequences of 4800 x 4800 (the batch-16 self-attention launch; rocprofv3 PMC passes of "
                                "scripts/pmc_attention.py on this kernel -- unchanged ISA apart from the optional log-sum-exp store since round 5); traffic_algorithmic = "
                                "q, k, v in and the output out, once",
                "pmc": "profiles/r5_pmc_attn32_v3.json"}
        variants.update(next_rows)
        if not args.no_match and extra and peaked is not None:
            q_ = peaked.get("q1", {})
            variants["peaked"] = {
                "workload": f"region B in the regime a trained matcher produces: NeRFMatchEvaluator.eval_data_loader, {Q} queries per batch, {peaked['steps']} timed "
                            f"batches, c2f matcher with `style=aligned` weights at temperature 30 and planted correspondences (image token i = code_i + noise, "
                            "point token i = rendered feature + code_i: one elementwise launch per batch inside the timed region); extract_matches.py:21-36 "
                            "returns ~1e3 matches on real data, the random-weight matcher of `query_images_per_sec` ~160",
                "query_images_per_sec": peaked["steps"] * world * Q / peaked["elapsed"], "ms_per_query": peaked["elapsed"] / (peaked["steps"] * Q) * 1e3,
                "matches_per_query": peaked.get("matches_per_query"), "matches_min": peaked.get("matches_min"), "matches_max": peaked.get("matches_max"),
                "latency_q1": {k: q_.get(k) for k in ("wall_ms", "wall_ms_p10", "wall_ms_p90", "gpu_ms", "native_calls", "matches", "fine_stage_ms", "loop_ms_per_query",
                                                      "spec_batches", "spec_reruns", "spec_cap", "loop_spec_batches", "loop_spec_reruns", "top_calls_ms")},
                "note": "spec_reruns = batches whose match count exceeded the speculative capacity of the single-pair path (the fine stage then runs a second "
                        "time behind the count read-back); fine_stage_ms = GPU span of the one nm_fine_stage launch of a one-query step"}
        if latency_q1 is not None:
            q16 = (elapsed_loc / (Ksteps * Q) * 1e3) if elapsed_loc else None
            c_ = latency_q1["c2f"]
            variants["latency_q1"] = {
                "workload": f"ONE {W}x{H} query per step, the reference's operating point (its eval loop is batch 1, nerfmatch_evaluator.py:631-724): "
                            f"NeRFMatchEvaluator.eval_batch = lean render_novel_view ({R} rays x ({S}+{S}) samples, {args.precision}) + NeRFMatcherMS.forward "
                            f"({R}x{R} tokens, mutual NN, fine stage, {mprec}); synchronize on both sides of every step, median of {c_['steps']} steps after warm-up; "
                            "backbone and PnP excluded",
                "wall_ms": c_["wall_ms"], "wall_ms_p10": c_["wall_ms_p10"], "wall_ms_p90": c_["wall_ms_p90"], "queries_per_s": 1e3 / c_["wall_ms"],
                "gpu_ms_native_calls": c_["gpu_ms"], "gpu_over_wall": c_["gpu_ms"] / c_["wall_ms"], "native_calls": c_["native_calls"],
                "gpu_note": "HIP events on the launch stream around every C-ABI call of a step, summed (a second pass: the event records never sit inside "
                            "the wall figure); exact kernel sums from the rocprofv3 trace: profiles/r5_latency_q1_*.json",
                "vs_q16_per_query": (c_["wall_ms"] / q16) if q16 else None, "q16_per_query_ms": q16,
                "loop_ms_per_query": latency_q1.get("loop_ms_per_query"),
                "loop_one_stream_ms_per_query": latency_q1.get("loop_one_stream_ms_per_query"),
                "loop_partitions": latency_q1.get("loop_partitions"),
                "loop_note": "NeRFMatchEvaluator.eval_data_loader over 40 batches of ONE query (the reference's loop as it is run: no synchronize between "
                             "steps beyond the matcher's own read-back), wall / 40, best of 3; N=1 only.  loop_ms_per_query = the shipped loop (round 6: query "
                             "i+1's render on a compute-unit partition beside query i's matcher on another), loop_one_stream_ms_per_query = the same "
                             "launches on one stream (round 5's loop); identical per-query results",
                "matches": c_["matches"], "top_calls_ms": c_["top_calls_ms"],
                "mini": {"workload": "the same step with the coarse-only model (NeRFMatcherCoarse: render + dual-softmax + mutual NN)",
                         **{k: latency_q1["coarse"][k] for k in ("wall_ms", "wall_ms_p10", "wall_ms_p90", "gpu_ms", "native_calls", "matches", "top_calls_ms")}}}
        if variants:
            line["variants"] = variants
        if mini is not None:
            line["mini"] = mini
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(S, sd if args.variant == "7scenes" else synth.nerf_state_dict(seed=0, density_bias=3.0))
    else:
        line = None
    if use_dist:
        dist.destroy_process_group()
    if line is not None:
        # last thing on stdout (RCCL writes its version banner straight to the descriptor when the group is created)
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
