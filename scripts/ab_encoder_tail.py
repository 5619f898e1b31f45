"""A/B of the fused encoder tail inside the c2f matcher forward (16 queries of 4800 x 4800 tokens, bf16x3), same process."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import ops, synth
from nerfmatch_amd.bench_match import build_evaluator
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
Q, H, W = 16, 480, 640
ev, make_batch = build_evaluator(dev, H, W, queries=Q)
nerfmatch_amd.set_precision("bf16x3")
R = 4800
g = torch.Generator().manual_seed(0)
base = dict(pt3d=torch.randn(Q, R, 3, generator=g).to(dev), pt_feat=torch.relu(torch.randn(Q, R, 256, generator=g)).to(dev) * 0.1,
            pt_mask=torch.ones(Q, R, dtype=torch.bool, device=dev))
ev.model.keep_conf = False
def run():
    b = make_batch(torch.eye(4)[None].repeat(Q, 1, 1), torch.eye(4))
    b.update(base)
    ev.model.forward(b, mutual=True)
    return b
for rep in range(2):
    for fused in (True, False):
        ops.ENCODER_TAIL_FUSED = fused
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print(f"fused tail {fused}: {e0.elapsed_time(e1) / 5:.3f} ms per matcher forward of {Q} queries")
