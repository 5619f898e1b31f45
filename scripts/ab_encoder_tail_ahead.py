"""A/B of the encoder tail's weight-stream depth (NM_TAIL_AHEAD = 3: rounds 3-4, 6: round 5): one process per library.
   NM_SRC=encoder_tail scripts/build_variants.sh "tail3:-DNM_TAIL_AHEAD=3" "tail6:-DNM_TAIL_AHEAD=6"; python scripts/ab_encoder_tail_ahead.py tail3 tail6"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
child = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from nerfmatch_amd import ops, synth
from nerfmatch_amd.modules.attention import GenericEncoderLayer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.LINEAR_PRECISION = "bf16x3"
layer = GenericEncoderLayer(model_dim=256, head_dim=32, att_mode="self")
sd = {}
synth._encoder_layer(sd, np.random.default_rng(3), "L", 256)
layer.load_state_dict({k[2:]: v for k, v in sd.items()}); layer.to(dev)
ff = layer.feedforward
out = []
for rows in (4800, 9600, 76800, 153600):
    att, xh = torch.randn(rows, 256, device=dev), torch.randn(rows, 256, device=dev)
    fn = lambda: ops.encoder_tail(att, xh, layer.attention.proj_out[0].weight, layer.norm2, ff.layers[0], ff.layers[2])
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        torch.cuda._sleep(int(4e6))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append(f"{rows}: {best:.1f} us")
print("  ".join(out))
''' % str(ROOT)
for rnd in range(2):
    for v in sys.argv[1:]:
        env = dict(os.environ, NERFMATCH_AMD_LIB=str(ROOT / "nerfmatch_amd/lib/variants" / f"lib_{v}.so"))
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        print(rnd, v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
