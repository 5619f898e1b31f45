"""Host / GPU profile of iNeRF steps with the matching term (where does a step's time go: torch profiler over 3 steps)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import inerf, latency, synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
K = synth.intrinsics(H, W)
img = torch.rand(H, W, 3, device=dev)
pose0 = torch.as_tensor(synth.camera_pose(1), dtype=torch.float32).to(dev)
nerfmatch_amd.set_precision("bf16x3")
ev, make_batch = build_evaluator(dev, H, W, queries=1)
R = (H // 8) * (W // 8)
match = dict(model=ev.model, image=torch.zeros(1, 3, H, W, device=dev), im_mask=torch.ones(1, R, dtype=torch.bool, device=dev),
             pt_mask=torch.ones(1, R, dtype=torch.bool, device=dev), unnorm=synth.unnorm_scene().to(dev))
inerf.refine(ren, K, H, W, img, pose0, num_optim=4, match=match)
torch.cuda.synchronize()
print("memory: allocated %.0f MB, reserved %.0f MB, num_alloc_retries %d, device mallocs %d" % (
    torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, torch.cuda.memory_stats()["num_alloc_retries"],
    torch.cuda.memory_stats()["num_device_alloc"]))
m0 = torch.cuda.memory_stats()["num_device_alloc"]
t0 = time.perf_counter()
inerf.refine(ren, K, H, W, img, pose0, num_optim=10, match=match)
torch.cuda.synchronize()
print("wall %.2f ms/step; device mallocs during the 10 steps: %d" % ((time.perf_counter() - t0) * 100, torch.cuda.memory_stats()["num_device_alloc"] - m0))
with latency.timed_lib() as tl:
    tl.spans = []
    inerf.refine(ren, K, H, W, img, pose0, num_optim=1, match=match)
    torch.cuda.synchronize()
    per = {}
    for name, e0, e1 in tl.spans:
        c = per.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += e0.elapsed_time(e1)
    print(f"one step: native calls {len(tl.spans)}, summed spans {sum(v[1] for v in per.values()):.2f} ms")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:40s} x{v[0]:3d} {v[1]:8.3f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    inerf.refine(ren, K, H, W, img, pose0, num_optim=3, match=match)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=70))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
