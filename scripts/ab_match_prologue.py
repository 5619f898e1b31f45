"""One-pair fused matching call (4800 x 4800): time per call by HIP events (used for the A/B of the merged prologue launches, round 5)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops, synth
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.MATCH_PRECISION = "bf16x3"
for P in (1, 16):
    a, b = synth.separated_features(4800, 4800, 256, seed=2)
    im, pt = a[None].expand(P, -1, -1).contiguous().to(dev), b[None].expand(P, -1, -1).contiguous().to(dev)
    for _ in range(5):
        r = ops.dual_softmax_match_batch(im, pt, 10.0, threshold=0.0, mutual=True, want_conf=False)
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = ops.dual_softmax_match_batch(im, pt, 10.0, threshold=0.0, mutual=True, want_conf=False)
    e1.record()
    torch.cuda.synchronize()
    k = int(r["count"][0])
    tail_zero = bool((r["i_ids"][0, k:] == 0).all() and (r["j_ids"][0, k:] == 0).all() and (r["mconf"][0, k:] == 0).all())
    print(f"P={P:2d}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call; matches {k}; slots behind the count are zeros: {tail_zero}")
