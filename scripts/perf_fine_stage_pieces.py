import sys
sys.path.insert(0, ".")
import torch, nerfmatch_amd
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
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, K in ((1, 192), (1, 512), (1, 4000)):
    ffeat = torch.randn(B, 128, 240, 320, generator=g).to(dev)
    flat = torch.sort(torch.randperm(B * 4800, generator=g)[:K]).values
    i_ids, map_ids = (flat % 4800).to(dev), (flat // 4800).to(dev)
    src = torch.randn(B * 4800, 256, generator=g).to(dev)
    ids = flat.to(dev)
    cnt = torch.tensor([K], dtype=torch.int32, device=dev)
    pf = ops.fine_pt_proj(src, ids, cnt, lin0, lin1)
    t_pp = bench(lambda: ops.fine_pt_proj(src, ids, cnt, lin0, lin1))
    t_l = bench(lambda: ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_f=pf))
    t_o = bench(lambda: ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4))
    t_1 = bench(lambda: ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_proj=(src, ids, lin0, lin1)))
    print(f"K={K}: pt_proj {t_pp:.1f} | layer with pt_f {t_l:.1f} | layer alone (output) {t_o:.1f} | one launch {t_1:.1f} us", flush=True)
