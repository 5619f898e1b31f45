"""A/B on one box: the fine stage's image side as one launch (ops.FINE_LAYER_FUSED) against window gather + generic layer kernels:
one-query steps (wall, native spans) and the 16-query loop."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from bench import Batches
from nerfmatch_amd import latency, ops, synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
unnorm = synth.unnorm_scene()
poses = [unnorm @ synth.camera_pose(seed=s) for s in range(64)]
ev, mk = build_evaluator(dev, 480, 640, queries=16)
kw = dict(renderer=ren, solver="none", query2query=True, mutual=True)
for rep in range(3):
    for flag in (True, False):
        setattr(ops, sys.argv[1] if len(sys.argv) > 1 else "FINE_LAYER_FUSED", flag)
        r = latency.measure(dev, ren, 480, 640, kind="c2f", n=40, queries=1, warmup=5)
        ev.eval_data_loader(data_loader=Batches(3, 0, 16, poses, unnorm, mk), **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev.eval_data_loader(data_loader=Batches(20, 3, 16, poses, unnorm, mk), **kw)
        torch.cuda.synchronize(); q16 = (time.perf_counter() - t0) / 320 * 1e3
        print(f"{sys.argv[1] if len(sys.argv) > 1 else 'FINE_LAYER_FUSED'}={flag}: one query wall {r['wall_ms']:.3f} ms (p10 {r['wall_ms_p10']:.3f}), native spans {r['gpu_ms']:.3f} ms, calls {r['native_calls']}; "
              f"16 per batch: {q16:.3f} ms per query")
