"""nm_fine_stage at sixteen peaked queries' matches (64000, sorted by (image, cell) as the matcher hands them over), a few launches, for rocprofv3 --pmc
passes of fine_layer_kernel (csrc/fine_layer.hip):   PMC_SCRIPT=pmc_fine_stage.py scripts/pmc_collect.sh fine_stage_r6"""
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
B, K = 16, 64000
ffeat = torch.randn(B, 128, 240, 320, generator=g).to(dev)
flat = torch.sort(torch.randperm(B * 4800, generator=g)[:K]).values
i_ids, map_ids = (flat % 4800).to(dev), (flat // 4800).to(dev)
src = torch.randn(B * 4800, 256, generator=g).to(dev)
ids = flat.to(dev)
cnt = torch.tensor([K], dtype=torch.int32, device=dev)
for _ in range(4):
    ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_proj=(src, ids, lin0, lin1))
torch.cuda.synchronize()
