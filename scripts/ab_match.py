"""Interleaved A/B timing of fused-matcher variants (scripts/build_variants.sh with NM_SRC=match_fused; one process per variant)."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
child = r'''
import sys, json, torch
sys.path.insert(0, %r)
from nerfmatch_amd import synth, ops
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.MATCH_PRECISION = "bf16x3"
res = {}
for T, P in ((4800, 16), (3600, 16)):
    g = torch.Generator().manual_seed(1)
    im = torch.randn(P, T, 256, generator=g).to(dev); pt = torch.randn(P, T, 256, generator=g).to(dev)
    f = lambda: ops.dual_softmax_match_batch(im, pt, 15.0, threshold=0.2, mutual=True, want_conf=False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3 / P)
    res[f"us_per_pair_{T}"] = round(best * 1e3, 2)
print(json.dumps(res))
''' % str(ROOT)
variants = sys.argv[1:]
for rnd in range(2):
    for v in variants:
        env = dict(os.environ, NERFMATCH_AMD_LIB=str(ROOT / "nerfmatch_amd/lib/variants" / f"lib_{v}.so"))
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
        print(rnd, v, line, flush=True)
