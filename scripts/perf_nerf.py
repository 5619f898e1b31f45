"""Timing of the fused NeRF pass, fp32-MFMA vs bf16x3-split kernels (HIP events on the launch stream)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import synth, ops
from nerfmatch_amd.nerf.renderer import NerfRenderer

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
FLOP_PER_SAMPLE = 1214464.0
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for S in (64, 128):
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
    ren.to(dev).eval()
    rays1, _ = ops.raygen(synth.intrinsics(), synth.camera_pose(1), 480, 640, dev)
    rays = rays1.repeat(Q, 1).contiguous()
    R = rays.shape[0]
    t = ops.sample_coarse(rays, torch.rand(R, S + 1, device=dev), S)
    for prec in ("fp32", "bf16x3"):
        blob = ren.nerf_fine.packed(dev, prec).clone()
        for name, kw in (("full", {}), ("no_rgb", dict(need_rgb=False))):
            for _ in range(3):
                ops.nerf_fwd(blob, rays, t, tap_layer=3, **kw)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.nerf_fwd(blob, rays, t, tap_layer=3, **kw)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            print(f"S={S} R={R} {prec:7s} {name:7s}: {best:.3f} ms/pass  {R*S/best*1e3/1e6:.1f} M ray-samples/s  {R*S*FLOP_PER_SAMPLE/best/1e9:.1f} TFLOP/s (fp32-equivalent algorithmic)")
