"""Time of nm_raygen_batch at 640 x 480 / ds 8 for 1 and 16 poses (NM_RAYGEN_PIXELS=0: the one-thread-per-ray kernel)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops, synth
dev = torch.device("cuda:0")
H, W = 480, 640
K = synth.intrinsics(H, W)
for Q in (1, 16):
    poses = torch.stack([synth.camera_pose(seed=s) for s in range(Q)])
    for _ in range(3):
        r, f = ops.raygen_batch(K, poses, H, W, dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        r, f = ops.raygen_batch(K, poses, H, W, dev)
    e1.record(); torch.cuda.synchronize()
    print(f"Q={Q:2d}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (memset + raygen + far fall-back); checksum {float(r.double().sum()):.8f}")
