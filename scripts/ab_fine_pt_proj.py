"""A/B on one box: the point side of the fine stage as one launch (ops.FINE_PT_PROJ_FUSED) against gather + two GEMM launches, one-query steps."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import latency, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
nerfmatch_amd.set_precision("bf16x3")
for rep in range(3):
    for flag in (True, False):
        ops.FINE_PT_PROJ_FUSED = flag
        r = latency.measure(dev, ren, 480, 640, kind="c2f", n=40, queries=1, warmup=5)
        print(f"fused={flag}: wall {r['wall_ms']:.3f} ms (p10 {r['wall_ms_p10']:.3f}), native spans {r['gpu_ms']:.3f} ms, calls {r['native_calls']}")
