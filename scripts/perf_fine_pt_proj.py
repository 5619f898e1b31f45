"""Time of nm_fine_pt_proj at one query's (512 slots, 190 valid) and sixteen queries' (3200) match counts."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
lin0, lin1 = torch.nn.Linear(256, 128).to(dev), torch.nn.Linear(128, 128).to(dev)
for K, count, rows in ((512, 190, 4800), (3200, 3200, 76800)):
    src = torch.randn(rows, 256, generator=g).to(dev)
    ids = torch.randint(0, rows, (K,), generator=g).to(dev)
    cnt = torch.tensor([count], dtype=torch.int32, device=dev)
    for _ in range(3): ops.fine_pt_proj(src, ids, cnt, lin0, lin1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): out = ops.fine_pt_proj(src, ids, cnt, lin0, lin1)
    e1.record(); torch.cuda.synchronize()
    print(f"K={K} count={count}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call; checksum {float(out.double().sum()):.6f}")
