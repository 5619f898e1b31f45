"""The whole fine stage in one launch (nm_fine_stage) at the match counts of the peaked regime: one query's ~4000 matches and sixteen queries' ~64000.
With NERFMATCH_AMD_LIB pointing at a -DFL_ABL=... variant library the output is a timing of the kernel without that phase (results meaningless)."""
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
lin0, lin1 = torch.nn.Linear(256, 128).to(dev), torch.nn.Linear(128, 128).to(dev)
def bench(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
row = []
for B, K in ((1, 192), (1, 4000), (16, 64000)):
    ffeat = torch.randn(B, 128, 240, 320, generator=g).to(dev)
    # matches as the matcher hands them over: sorted by (image, cell); at K >= 4000 per image nearly every cell is matched
    flat = torch.sort(torch.randperm(B * 4800, generator=g)[:K] if K <= B * 4800 else torch.randint(0, B * 4800, (K,), generator=g)).values
    i_ids, map_ids = (flat % 4800).to(dev), (flat // 4800).to(dev)
    src = torch.randn(B * 4800, 256, generator=g).to(dev)
    ids = (map_ids * 4800 + torch.randint(0, 4800, (K,), generator=g).to(dev))
    cnt = torch.tensor([K], dtype=torch.int32, device=dev)
    t = bench(lambda: ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_proj=(src, ids, lin0, lin1)))
    def two():
        pf = ops.fine_pt_proj(src, ids, cnt, lin0, lin1)
        return ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_f=pf)
    def gemms():
        pf = ops.gather_rows(src, ids, cnt)
        pf = ops.linear(pf, lin0.weight, lin0.bias)
        pf = ops.linear(pf, lin1.weight, lin1.bias)
        return ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_f=pf)
    t2, t3 = bench(two), bench(gemms)
    a, b, c = ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_proj=(src, ids, lin0, lin1)), two(), gemms()
    row.append(f"K={K}: one launch {t:.1f} us | pt_proj launch + layer {t2:.1f} (max diff {float((a - b).abs().max()):.1e}) | gather + 2 GEMMs + layer {t3:.1f} ({float((a - c).abs().max()):.1e})")
print("\n".join(row))
