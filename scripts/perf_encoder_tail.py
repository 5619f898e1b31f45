"""Fused encoder tail against the four launches it replaces, at the evaluator's row counts (32 sequences x 4800 tokens)."""
import sys, time
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
for rows in (76800, 153600):
    att, xh = torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev)
    def fused():
        return ops.encoder_tail(att, xh, layer.attention.proj_out[0].weight, layer.norm2, ff.layers[0], ff.layers[2])
    def separate():
        a = ops.linear(att, layer.attention.proj_out[0].weight, residual=xh)
        a = ops.layernorm(a, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        return ff(a, residual=xh)
    for name, fn in (("fused", fused), ("separate", separate)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"rows {rows}: {name:9s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
