"""NerfEvaluator.cache_scene_pts: frames per second with 0 (serial: render, read back, write), 1, 2, 3 writer threads.

    python scripts/perf_cache_frames.py [frames]
"""
import sys, tempfile, time
from argparse import Namespace
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import ops, synth
from nerfmatch_amd.nerf_evaluator import NerfEvaluator

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
H, W, DS, S = 480, 640, 8, 64
R = (H // DS) * (W // DS)
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = synth.nerf_config("7scenes", num_pts=S, img_wh=(W, H))
cfg.exp, cfg.split, cfg.downsample = Namespace(seed=0), "train", DS
kmat, unnorm = synth.intrinsics(H, W), synth.unnorm_scene()
frames = []
for f in range(nfr):
    rays, _ = ops.raygen(kmat, synth.camera_pose(f), H, W, dev)
    frames.append(dict(img_wh=torch.tensor([[W // DS, H // DS]]), rays=rays[None], rgbs=torch.zeros(1, R, 3), img_idx=[f"seq1_frame{f:05d}"], unnorm_scene=unnorm[None]))
ev = NerfEvaluator(cfg, vocab_num=8, stop_layer=3, data_loader=frames)
ev.model.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0), strict=True)
ev.model.precision = "fp16x3"
with tempfile.TemporaryDirectory() as td:
    ev.cache_scene_pts(cache_dir=Path(td) / "warm", frames_per_launch=4)
    for writers in (1, 2, 3, 1, 2):
        ev.cache_writers = writers
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.cache_scene_pts(cache_dir=Path(td) / f"w{writers}", frames_per_launch=4)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{writers} writer thread(s): {nfr / el:7.1f} frames/s ({el / nfr * 1e3:.2f} ms per frame)", flush=True)
    for fpl in (8, 16):
        ev.cache_writers = 2
        ev.cache_scene_pts(cache_dir=Path(td) / f"warm{fpl}", frames_per_launch=fpl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev.cache_scene_pts(cache_dir=Path(td) / f"f{fpl}", frames_per_launch=fpl)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"2 writers, {fpl} frames per launch: {nfr / el:7.1f} frames/s", flush=True)
