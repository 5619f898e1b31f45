"""Time of the fine stage's image side: nm_fine_window_layer (one launch) against window gather + the generic layer kernels (bf16x3), at one
query's and sixteen queries' match counts."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import nerfmatch_amd
from nerfmatch_amd import ops
from nerfmatch_amd.modules.attention import SelfAttentionBlock
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
nerfmatch_amd.set_precision("bf16x3")
g = torch.Generator().manual_seed(0)
block = SelfAttentionBlock(1, 128, att_type="full", head_dim=16).to(dev).eval()
def bench(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, K, count in ((1, 512, 190), (16, 3200, 3200), (16, 4096, 3200)):
    ffeat = torch.randn(B, 128, 240, 320, generator=g).to(dev)
    i_ids = torch.randint(0, 4800, (K,), generator=g).to(dev)
    map_ids = torch.randint(0, B, (K,), generator=g).to(dev)
    cnt = torch.tensor([count], dtype=torch.int32, device=dev)
    t_f = bench(lambda: ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4))
    t_g = bench(lambda: block(ops.fine_windows_batch(ffeat, map_ids, i_ids, cnt, 5, 4)))
    print(f"K={K} count={count}: one launch {t_f:.1f} us; gather + generic layer {t_g:.1f} us")
