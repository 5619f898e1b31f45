"""Interleaved A/B timing of nerf_fwd variants (one process per variant because the library is loaded once)."""
import os, subprocess, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
child = r'''
import sys, json, torch
sys.path.insert(0, %r)
from nerfmatch_amd import synth, ops
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
res = {}
for S in (64,):
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
    rays = torch.cat([ops.raygen(synth.intrinsics(), synth.camera_pose(q), 480, 640, dev)[0] for q in range(4)])  # 4 queries per launch
    t = ops.sample_coarse(rays, torch.rand(rays.shape[0], S + 1, device=dev), S)
    import os
    blob = ren.nerf_fine.packed(dev, os.environ.get("NM_PRECISION", "fp32"))
    if hasattr(blob, "nm_guard"): pass
    for name, kw in (("full", dict()), ("nofeat", dict(need_feat=False)), ("density", dict(need_feat=False, need_rgb=False))):
        for _ in range(3): ops.nerf_fwd(blob, rays, t, tap_layer=3, **kw)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.nerf_fwd(blob, rays, t, tap_layer=3, **kw)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        res[f"{S}_{name}"] = round(best, 4)
print(json.dumps(res))
''' % str(ROOT)
variants = sys.argv[1:]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ, NERFMATCH_AMD_LIB=str(ROOT / "nerfmatch_amd/lib/variants" / f"lib_{v}.so"))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]
        print(rnd, v, line, flush=True)
