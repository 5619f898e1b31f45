"""Workload for scripts/pmc_collect.sh (PMC_SCRIPT=pmc_encoder_tail.py): the fused encoder tail at 32 sequences x 4800 tokens."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from nerfmatch_amd import ops, synth
from nerfmatch_amd.modules.attention import GenericEncoderLayer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.LINEAR_PRECISION = "bf16x3"
layer = GenericEncoderLayer(model_dim=256, head_dim=32, att_mode="self")
sd = {}
synth._encoder_layer(sd, np.random.default_rng(3), "L", 256)
layer.load_state_dict({k[2:]: v for k, v in sd.items()})
layer.to(dev)
ff = layer.feedforward
rows = 153600
att, xh = torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev)
for _ in range(6):
    ops.encoder_tail(att, xh, layer.attention.proj_out[0].weight, layer.norm2, ff.layers[0], ff.layers[2])
torch.cuda.synchronize()
