"""Per-phase timeline of the bf16x3 fused NeRF kernel from an -DNM_TRACE build (s_memtime stamps of wavefront 0).

  NM_SRC=nerf_fwd_bf16 scripts/build_variants.sh "trace:-DNM_TRACE=1"
  NERFMATCH_AMD_LIB=nerfmatch_amd/lib/variants/lib_trace.so python scripts/trace_nerf.py
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from nerfmatch_amd import synth, ops
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
S = 64
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
rays, _ = ops.raygen(synth.intrinsics(), synth.camera_pose(1), 480, 640, dev)
t = ops.sample_coarse(rays, torch.rand(rays.shape[0], S + 1, device=dev), S)
import os
blob = ren.nerf_fine.packed(dev, os.environ.get("NM_TRACE_PREC", "bf16x3"))
for _ in range(3):
    out = ops.nerf_fwd(blob, rays, t, tap_layer=3, want_raw=True, need_rgb=os.environ.get("NM_TRACE_RGB", "1") == "1", need_feat=os.environ.get("NM_TRACE_FEAT", "1") == "1")
torch.cuda.synchronize()
nblk = rays.shape[0] * S // 128
raw = out["raw"].cpu().numpy().view(np.uint64).reshape(-1)[: nblk * 32].reshape(nblk, 32).astype(np.int64)
names = ["small copy", "ray/t loads + IPE"] + [f"layer {l}" for l in range(8)] + ["views K-loop", "rgb head", "barrier", "composite+sums", "feature reduce", "barrier", "feat combine + stores"]
d = np.diff(raw[:, :18], axis=1)
print(f"blocks {nblk}; s_memtime ticks (100 MHz ref => x{2.2e9/1e8:.0f} shader cycles if the counter is REFCLK)")
tot = raw[:, 17] - raw[:, 0]
print(f"tile total: median {np.median(tot):.0f}  p10 {np.percentile(tot,10):.0f}  p90 {np.percentile(tot,90):.0f} ticks")
for i, n in enumerate(names):
    print(f"  {n:24s} median {np.median(d[:, i]):9.0f}   ({100*np.median(d[:, i])/np.median(tot):5.1f} %)")
span = raw[:, 17].max() - raw[:, 0].min()
print(f"kernel span {span} ticks; sum of tiles / 256 CUs = {tot.sum()/256:.0f}")
# gap between consecutive tiles on the same CU is not observable without HW_ID; report start-time histogram instead
st = np.sort(raw[:, 0] - raw[:, 0].min())
print("tile start times (ticks), every 256th:", st[::256])
