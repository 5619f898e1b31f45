#!/usr/bin/env python3
"""Benchmark of the NeRFMatch hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one batch of Q query images on every rank (--queries, default 16; the reference's eval loop takes the
batch size as an argument, nerfmatch_evaluator.py:726-731,864-869, default 1).  Two timed regions of EXACTLY K steps
each (SURVEY.md section 8d defines two metrics):
  A  render_novel_views of Q 640x480 queries at downsample 8: Q x 4800 rays x (S+S) samples through the coarse and
     fine NeRF, fp32, all reference outputs -> `value` = rays*samples/sec (whole job);
  B  the same render followed by the coarse-to-fine 2D-3D match against the rendered points (image backbone
     excluded, PnP excluded) -> `query_images_per_sec`.
Queries shard over ranks with no data-path collective; the per-query pose-candidate records are all-gathered
once at the end of each region (RCCL over xGMI), inside the timed region.

Prints ONE JSON line on rank 0 (see the task contract): metric rays*samples/sec (whole job), plus
  roofline     : dominant kernel (nerf_fwd_kernel) FLOP/launch / its mean duration measured with HIP events on
                 the launch stream inside the timed region, against the 157.3 TFLOP/s fp32-MFMA peak;
  cpu_baseline : the oracle (CPU restatement of the reference, torch-CPU fp32, all host cores) timed on a bounded
                 sample of the same workload (rank 0, N=1 only).
"""
import argparse
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

FLOP_PER_SAMPLE_PASS = 1_214_464  # 2 x 607,232 MAC: SURVEY.md section 8d (7-Scenes config, no appearance embedding)
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0}  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / dense bf16 MFMA peaks
KERNEL = {"fp32": "nerf_fwd_kernel", "bf16x3": "nerf_fwd_bf16x3_kernel"}
# MFMA FLOPs the bf16x3 kernel EXECUTES per algorithmic FLOP: 3 products per fp32 product, K padded 90->96 / 27+16->48
BF16X3_EXEC_FLOP_PER_SAMPLE_PASS = 3 * 2 * (96 * 256 + 4 * 65536 + (96 + 256) * 256 + 2 * 65536 + 65536 + (256 + 48) * 128)
H, W, DS = 480, 640, 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=64, help="samples per ray per pass (coarse and fine)")
    ap.add_argument("--queries", type=int, default=16, help="query images per step per GPU (the reference's eval batch_size; 1 = its default)")
    ap.add_argument("--precision", choices=["bf16x3", "fp32"], default="bf16x3",
                    help="matrix-core arithmetic of the fused NeRF kernel: bf16x3 = bf16 MFMA with fp32-accurate hi/lo operand "
                         "splitting (default; < 1e-6 from the fp32 path), fp32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true", help="render only")
    return ap.parse_args()


def cpu_baseline(S, seed_sd):
    """Oracle render of a bounded sample (1200 of the 4800 rays x (S+S) samples).  The thread count is chosen by a
    short sweep (more threads than physical cores available to the container only slows torch-CPU down); the
    reported value is the median of 3 runs after 1 warm-up at the best setting."""
    from nerfmatch_amd import synth
    from oracle import nerf_oracle as no

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    K = synth.intrinsics(H, W)
    rays = no.make_rays(H, W, K, synth.camera_pose(1), ds=DS)[::4].contiguous()
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
        dt = run(rays[:300], t_rand[:300], jit[:300])
        if dt < best_t:
            best_n, best_t = n, dt
        if dt > 2.0 * best_t:
            break
    torch.set_num_threads(best_n)
    times = [run(rays, t_rand, jit) for _ in range(4)]
    med = statistics.median(times[1:])
    return dict(value=R * 2 * S / med, unit="rays*samples/s", cores=best_n, kind="port",
                sample=f"oracle.render_rays on {R} of 4800 rays x ({S}+{S}) samples, median of 3 runs after 1 warm-up, "
                       f"torch-CPU fp32, {best_n} threads (best of a sweep; {avail} logical CPUs visible)")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    torch.set_grad_enabled(False)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # NM_FORCE_DIST=1 exercises the RCCL code path (init, barrier, all_gather, all_reduce) even at world size 1
    use_dist = world > 1 or os.environ.get("NM_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    from nerfmatch_amd import synth, ops
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    S = args.samples
    cfg = synth.nerf_config("7scenes", num_pts=S)
    sd = synth.nerf_state_dict(seed=0, density_bias=3.0)
    ren = NerfRenderer(cfg, training=False, stop_layer=3)
    ren.load_state_dict(sd)
    ren.to(dev).eval()
    ren.precision = args.precision
    K = synth.intrinsics(H, W)
    unnorm = synth.unnorm_scene()
    R = (H // DS) * (W // DS)

    matcher = None
    if not args.no_match:
        try:
            from nerfmatch_amd.bench_match import build_matcher  # provided once the matcher kernels exist
            matcher = build_matcher(dev, H, W, queries=args.queries)
        except ImportError:
            matcher = None

    # instrument the dominant kernel: HIP events around every nm_nerf_fwd launch (same stream as the launches)
    kernel_events = []
    raw_fwd = ops.nerf_fwd

    def timed_fwd(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = raw_fwd(*a, **kw)
        e1.record()
        # samples the launch really evaluates: all R*S, or R*(S/2+1) when the bf16x3 kernel skips the zero-width tail of
        # the fine pass (NM_NERF_ZERO_TAIL; the skipped samples have weight exactly 0 in every output)
        rr, ss = a[2].shape[0], a[2].shape[1] - 1
        skip = kw.get("zero_tail", False) and a[0].dtype == torch.uint8 and (ss in (64, 128) or ss % 256 == 0) and not kw.get("want_raw") and not kw.get("want_sample_feat") and not kw.get("feat_max")
        kernel_events.append((e0, e1, rr * (ss // 2 + 1) if skip else rr * ss))
        return out

    n_rec = (args.steps + args.warmup) * args.queries

    Q = args.queries
    poses = [unnorm @ synth.camera_pose(seed=s_) for s_ in range(64)]

    def make_step(with_match, records):
        """step(i) issues query batch i.  In the localisation region the steps are software pipelined on the host: the
        matcher of batch i is enqueued up to its one synchronisation point (match-count read-back), then batch i+1's render
        is issued BEFORE that read-back, so the GPU has work queued while the host waits and then issues the fine stage.
        step.flush() completes the batch still in flight."""
        pending = []

        def finish_pending():
            while pending:
                st, i0 = pending.pop()
                records[i0 * Q:(i0 + 1) * Q, 18] = matcher.finish(st)

        def step(i):
            # global query indices of this step: batches of Q consecutive queries, round-robin over ranks
            q0 = (i * world + rank) * Q
            c2ws = torch.stack([poses[(q0 + j) % 64] for j in range(Q)])
            # region A computes every output the reference's render_rays returns; the localisation region renders what the
            # evaluator's loop reads (pt3d, pt_feat: nerfmatch_evaluator.py:566-573): the coarse pass keeps only the density
            # head and the fine pass skips feature_linear / views / rgb
            out = ren.render_novel_views((H, W), K, c2ws, unnorm, dev, lean=with_match, want_im_pred=not with_match)
            rec = records[i * Q:(i + 1) * Q]
            rec[:, 0] = torch.arange(q0, q0 + Q, device=dev)
            rec[:, 1:17] = c2ws.reshape(Q, 16).to(dev, non_blocking=True)
            rec[:, 17] = out["pt_feat"][:, 0, 0]
            if with_match:
                st = matcher.begin(out)
                finish_pending()          # batch i-1: count read-back + fine stage, behind batch i's queued work
                pending.append((st, i))

        step.flush = finish_pending
        return step

    def timed_region(with_match):
        """W warm-up steps, then EXACTLY K steps between barrier+synchronize brackets; the shard's pose-candidate
        records are all-gathered (RCCL) inside the region.  Returns the max-over-ranks wall time."""
        records = torch.zeros(n_rec, 20, device=dev)
        step = make_step(with_match, records)
        for i in range(args.warmup):
            step(i)
        step.flush()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        step.flush()  # the last batch's fine stage belongs to the timed region
        if use_dist:
            gathered = [torch.empty_like(records) for _ in range(world)]
            dist.all_gather(gathered, records)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    # region A (metric i, `value`): render only, with the dominant kernel instrumented
    import nerfmatch_amd.nerf.renderer as rmod
    ops.nerf_fwd = timed_fwd
    rmod.ops.nerf_fwd = timed_fwd
    elapsed = timed_region(False)
    main_events, kernel_events = kernel_events, []
    # the other arithmetic path, same region definition (reported as extra fields, not as `value`)
    other = "fp32" if args.precision == "bf16x3" else "bf16x3"
    ren.precision = other
    elapsed_other = timed_region(False)
    other_events = kernel_events
    kernel_events = []
    ren.precision = args.precision
    # the same region with every sample evaluated (no zero-tail skip): reported beside `value`, so that the effect of
    # skipping the provably zero-weight fine samples is visible in the line itself
    elapsed_full = None
    if args.precision == "bf16x3" and ren.skip_zero_tail:
        ren.skip_zero_tail = False
        elapsed_full = timed_region(False)
        ren.skip_zero_tail = True
    kernel_events = main_events
    ops.nerf_fwd = raw_fwd
    rmod.ops.nerf_fwd = raw_fwd
    # region B (metric ii): full localisation step = render + coarse-to-fine match
    ops.ATTENTION_PRECISION = args.precision  # the matcher's contractions (attention, nn.Linear) follow the same arithmetic choice
    ops.LINEAR_PRECISION = args.precision
    ops.MATCH_PRECISION = args.precision
    elapsed_loc = timed_region(True) if matcher is not None else None
    ops.ATTENTION_PRECISION = "fp32"
    ops.LINEAR_PRECISION = "fp32"
    ops.MATCH_PRECISION = "fp32"

    kern_ms = [a.elapsed_time(b) for a, b, _ in kernel_events]
    kern_samples = [n for _, _, n in kernel_events]
    # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (profiles/r1_pmc_nerf_fwd.json,
    # measured at R=4800,S=64 per launch); it scales with the ray count, so it is reported per launch of Q*R rays.
    traffic = None
    pmc = ROOT / "profiles" / "r1_pmc_nerf_fwd.json"
    pmc = ROOT / "profiles" / ("r1_pmc_nerf_fwd.json" if args.precision == "fp32" else "r1_pmc_nerf_fwd_bf16x3.json")
    if pmc.exists() and S == 64:
        traffic = json.load(open(pmc))["derived"]["traffic_bytes"] * args.queries
    if rank == 0:
        total_units = world * args.steps * Q * R * 2 * S
        avg_kernel_s = (sum(kern_ms) / len(kern_ms)) * 1e-3
        flop_per_launch = Q * R * S * FLOP_PER_SAMPLE_PASS                       # the reference's arithmetic for one pass
        evaluated_per_launch = sum(kern_samples) / len(kern_samples)           # samples the kernel really runs the MLP on
        achieved = evaluated_per_launch * FLOP_PER_SAMPLE_PASS / avg_kernel_s / 1e12
        peak = PEAK_TFLOPS[args.precision]
        other_ms = sum(a.elapsed_time(b) for a, b, _ in other_events) / len(other_events)
        other_samples = sum(n for _, _, n in other_events) / len(other_events)
        line = {
            "metric": "rays*samples/sec",
            "value": total_units / elapsed,
            "unit": "rays*samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16x3 (bf16 MFMA on hi/lo-split fp32 operands, fp32 accumulate; fp32 everywhere else)" if args.precision == "bf16x3" else "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{Q} 7-Scenes-style queries per rank and step (one batch): render_novel_views 640x480 ds8 -> {Q}x{R} rays x ({S}+{S}) samples "
                            f"(coarse+fine 8x256 NeRF, stop_layer 3, ret_pfeat, all reference outputs, {args.precision} kernel"
                            + ("; the fine pass runs the MLP on samples 0..S/2 only: the reference's randomized resampler leaves the other intervals with zero width = weight exactly 0, outputs identical" if args.precision == "bf16x3" else "")
                            + ") [timed region of `value`]; "
                            f"query_images_per_sec = a second timed region of the same K steps: render of pt3d / pt_feat only (no colour heads, as the evaluator's loop reads them) + the c2f matcher, "
                            f"host-pipelined so that batch i+1's render is enqueued before batch i's match-count read-back "
                            f"({R}x{R} tokens, mutual NN, fine stage; image backbone excluded) appended to every step",
                "rays": R, "samples_coarse": S, "samples_fine": S, "queries_per_step_per_gpu": Q,
                "sharding": "query images round-robin over ranks; one all_gather of pose-candidate records at shard end",
            },
            "full_evaluation": None if elapsed_full is None else {
                "value": total_units / elapsed_full, "ms_per_step": elapsed_full / args.steps * 1e3,
                "note": "same region A with NM_NERF_ZERO_TAIL off: the fine pass runs the MLP on all S samples like the reference "
                        "(the samples `value` skips have zero interval width, i.e. weight exactly 0 in every output)"},
            "query_images_per_sec": (world * args.steps * Q / elapsed_loc) if elapsed_loc else None,
            "localize_ms_per_query": (elapsed_loc / (args.steps * Q) * 1e3) if elapsed_loc else None,
            "roofline": {
                "bound": "mfma", "kernel": KERNEL[args.precision], "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                "achieved_note": "fp32-equivalent FLOP of the samples a launch EVALUATES (1,214,464 each; the fine pass skips the "
                                 "zero-width tail the reference's resampler produces: S/2+1 of S samples, identical outputs) / mean launch duration",
                "evaluated_samples_per_launch": evaluated_per_launch, "samples_per_launch_incl_skipped": Q * R * S,
                "traffic_note": "L2<->fabric bytes per launch from rocprofv3 PMC passes (profiles/r1_pmc_nerf_fwd*.json), FETCH_SIZE x2-corrected + WRITE_SIZE, scaled by the ray count; for the bf16x3 kernel this is the round trip of the tapped layer-3 activations through its 64 MiB workspace (128 KiB per 128-sample tile each way, served by the Infinity Cache: the counters sit in front of it), not re-reads of inputs: algorithmic bytes are ~55 B per ray in and ~1.1 KB per ray out",
                "flop_per_launch": flop_per_launch, "avg_launch_ms": avg_kernel_s * 1e3, "launches_timed": len(kern_ms),
            },
        }
        if args.precision == "bf16x3":
            ex = evaluated_per_launch * BF16X3_EXEC_FLOP_PER_SAMPLE_PASS / avg_kernel_s / 1e12
            line["roofline"].update(executed_mfma_tflops=ex, frac_executed=ex / peak,
                                    executed_note="bf16 MFMA FLOP actually issued: 3 per fp32 product (w_hi*x_hi + w_hi*x_lo + w_lo*x_hi), padded K")
        line["other_precision"] = {
            "precision": other, "kernel": KERNEL[other], "value": total_units / elapsed_other, "unit": "rays*samples/s",
            "avg_launch_ms": other_ms, "achieved_tflops": other_samples * FLOP_PER_SAMPLE_PASS / (other_ms * 1e-3) / 1e12,
            "frac_of_peak": other_samples * FLOP_PER_SAMPLE_PASS / (other_ms * 1e-3) / 1e12 / PEAK_TFLOPS[other], "peak": PEAK_TFLOPS[other],
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(S, sd)
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
