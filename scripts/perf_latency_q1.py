"""The reference's own operating point: ONE query per step (nerfmatch_evaluator.py:631-724 loops with batch size 1 and prints
"Avg match time" per query, utils/metrics.py:589-593).  Wall time of one localisation step = NeRFMatchEvaluator.eval_batch
(lean render_novel_view + NeRFMatcherMS.forward / NeRFMatcherCoarse.forward, solver none) with a synchronize on both sides,
beside the GPU time of its native calls (HIP events around every C-ABI launch) and their count.

    python scripts/perf_latency_q1.py [c2f|coarse] [n] [Q]
Under `rocprofv3 --kernel-trace` the last n steps' kernels give the exact kernel-time sum (scripts/latency_trace_summarize.py)."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import nerfmatch_amd
from nerfmatch_amd import latency, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer

kind = sys.argv[1] if len(sys.argv) > 1 else "c2f"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W = 480, 640
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
res = latency.measure(dev, ren, H, W, kind=kind, n=n, queries=Q, warmup=5, gap_s=float(os.environ.get('NM_LAT_GAP_MS', '0')) * 1e-3)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items() if k not in ("per_call", "series")})
for name, (cnt, ms) in sorted(res["per_call"].items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:36s} x{cnt:5.1f}  {ms:8.4f} ms")
