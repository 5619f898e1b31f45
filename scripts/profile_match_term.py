"""Where the matcher's share of an iNeRF step with the matching term goes (inerf._match_term = NeRFMatcherMS.match_loss + autograd.grad w.r.t.
pt_feat / pt3d at 4800 x 4800 tokens): wall per call, GPU spans of its native calls, their count, host profile."""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import inerf, latency, synth
from nerfmatch_amd.bench_match import build_evaluator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
nerfmatch_amd.set_precision("bf16x3")
ev, make_batch = build_evaluator(dev, H, W, queries=1)
R = (H // 8) * (W // 8)
match = dict(model=ev.model, image=torch.zeros(1, 3, H, W, device=dev), im_mask=torch.ones(1, R, dtype=torch.bool, device=dev),
             pt_mask=torch.ones(1, R, dtype=torch.bool, device=dev), unnorm=synth.unnorm_scene().to(dev))
g = torch.Generator().manual_seed(0)
pt_feat = torch.relu(torch.randn(R, 256, generator=g)).to(dev)
pt3d = (torch.randn(R, 3, generator=g) * 0.25).to(dev)
for _ in range(3):
    inerf._match_term(match, pt_feat, pt3d)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    inerf._match_term(match, pt_feat, pt3d)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / n * 1e3:.2f} ms per call (back to back)")
with latency.timed_lib() as tl:
    tl.spans = []
    inerf._match_term(match, pt_feat, pt3d)
    torch.cuda.synchronize()
    per = {}
    for name, e0, e1 in tl.spans:
        c = per.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += e0.elapsed_time(e1)
    print(f"native calls {len(tl.spans)}, summed spans {sum(v[1] for v in per.values()):.2f} ms")
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"   {k:40s} x{v[0]:3d} {v[1]:8.3f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    inerf._match_term(match, pt_feat, pt3d)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    inerf._match_term(match, pt_feat, pt3d)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
