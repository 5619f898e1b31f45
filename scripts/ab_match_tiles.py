"""A/B of the fused matching's tile width: 16 pairs of 4800 x 4800 (and 3600 x 3600) tokens per call, time per call by HIP events.
    python scripts/ab_match_tiles.py          (256-column tiles, round 5)
    NM_MATCH_NB=4 python scripts/ab_match_tiles.py   (128 x 128 tiles of rounds 3-4)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.MATCH_PRECISION = "bf16x3"
print("NM_MATCH_NB =", os.environ.get("NM_MATCH_NB", "8"))
for T in (4800, 3600):
    for P in (16, 1):
        a, b = synth.separated_features(T, T, 256, seed=2)
        im, pt = a[None].expand(P, -1, -1).contiguous().to(dev), b[None].expand(P, -1, -1).contiguous().to(dev)
        for _ in range(3):
            r = ops.dual_softmax_match_batch(im, pt, 10.0, threshold=0.0, mutual=True, want_conf=False)
        torch.cuda.synchronize()
        n = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            r = ops.dual_softmax_match_batch(im, pt, 10.0, threshold=0.0, mutual=True, want_conf=False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"T={T} P={P:2d}: {ms:.3f} ms per call = {ms / P * 1e3:.1f} us per pair, {2.0 * T * T * 256 * P / ms / 1e9:.0f} TFLOP/s algorithmic "
              f"({2.0 * T * T * 256 * P / ms / 1e9 / 2500:.3f} of 2.5 PF); matches in pair 0: {int(r['count'][0])}")
