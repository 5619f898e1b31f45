"""Quick timing of the fused NeRF pass (HIP events on the launch stream)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import synth, ops
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
FLOP_PER_SAMPLE = 1214464.0
for S in (64, 128):
    cfg = synth.nerf_config("7scenes", num_pts=S)
    ren = NerfRenderer(cfg, training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
    ren.to(dev).eval()
    rays, _ = ops.raygen(synth.intrinsics(), synth.camera_pose(1), 480, 640, dev)
    R = rays.shape[0]
    t = ops.sample_coarse(rays, torch.rand(R, S + 1, device=dev), S)
    blob = ren.nerf_fine.packed(dev)
    for name, kw in (("full", {}), ("no_rgb", dict(need_rgb=False)), ("sigma_only", dict(need_rgb=False, need_feat=False))):
        for _ in range(3):
            ops.nerf_fwd(blob, rays, t, tap_layer=3, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.nerf_fwd(blob, rays, t, tap_layer=3, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"S={S} R={R} {name}: {ms:.3f} ms/pass  {R*S/ms*1e3/1e6:.1f} M ray-samples/s  {R*S*FLOP_PER_SAMPLE/ms/1e9:.1f} TFLOP/s(full-count)")
    ren.ret_pfeat = True
    for lean in (False, True):
        for _ in range(2):
            ren.render_novel_view((480, 640), synth.intrinsics(), synth.camera_pose(1), torch.eye(4), dev, lean=lean)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ren.render_novel_view((480, 640), synth.intrinsics(), synth.camera_pose(1), torch.eye(4), dev, lean=lean)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"S={S} render_novel_view lean={lean}: {ms:.3f} ms  {R*2*S/ms*1e3/1e6:.1f} M ray-samples/s")
