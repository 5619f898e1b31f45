"""Aggregate accuracy of the fused passes on the ten trained-like fixtures (tests/golden/nerf_surf_w*_p*.npz): rms and max of
|HIP - reference fp32|, |HIP - reference fp64| and |reference fp32 - fp64| over all rays of all fixtures, per output.
One process per library variant (NERFMATCH_AMD_LIB), e.g.
    python scripts/surface_seed_stats.py fp32 fp16x3 fp16x3:neutral
`:neutral` skips the activation-scale calibration (weights still scaled unless the library was built with -DNM_NO_WSCALE)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch

from conftest import load_golden
from test_nerf_gpu import make_renderer
from test_surface_seeds_gpu import SEEDS, REF_KEY

torch.set_grad_enabled(False)
gpu = torch.device("cuda:0")
KEYS = ("weights_coarse", "feat_coarse", "weights_fine", "feat_fine")
for spec in sys.argv[1:] or ["fp32", "fp16x3"]:
    precision, _, opt = spec.partition(":")
    acc = {k: dict(hr=[], ht=[], rt=[]) for k in KEYS}
    scales = None
    for ws, ps in SEEDS:
        fx = load_golden(f"nerf_surf_w{ws}_p{ps}")
        ren, sd = make_renderer(fx, gpu)
        rays = fx["rays"].to(gpu)
        if opt == "neutral":
            from nerfmatch_amd import ops
            oc = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, precision), rays, fx["t_coarse"].to(gpu), None, tap_layer=-1)
            of = ops.nerf_fwd(ren.nerf_fine.packed(gpu, precision), rays, fx["t_fine"].to(gpu), None, tap_layer=3)
        else:
            oc = ren.nerf_coarse.fused(precision, rays, fx["t_coarse"].to(gpu), None, tap_layer=-1)
            of = ren.nerf_fine.fused(precision, rays, fx["t_fine"].to(gpu), None, tap_layer=3)
            if precision == "fp16x3":
                scales = (ren.nerf_coarse._act_log2.get(str(gpu)), ren.nerf_fine._act_log2.get(str(gpu)))
        got = {"weights_coarse": oc["weights"], "feat_coarse": oc["feat"], "weights_fine": of["weights"], "feat_fine": of["feat"]}
        for k in KEYS:
            ref = fx[REF_KEY.get(k, f"pred_{k}")].double()
            truth = ref + fx[f"truth_d_{k}"].double()
            g = got[k].cpu().double()
            acc[k]["hr"].append((g - ref).flatten()); acc[k]["ht"].append((g - truth).flatten()); acc[k]["rt"].append((ref - truth).flatten())
    print(f"== {spec}" + (f"  (last calibrated scales coarse {scales[0]} fine {scales[1]})" if scales else ""))
    for k in KEYS:
        c = {n: torch.cat(v) for n, v in acc[k].items()}
        rms = lambda x: float(x.pow(2).mean().sqrt())
        print(f"  {k:15s} rms |hip-ref| {rms(c['hr']):.2e} |hip-fp64| {rms(c['ht']):.2e} |ref-fp64| {rms(c['rt']):.2e}   "
              f"max |hip-ref| {float(c['hr'].abs().max()):.2e} |hip-fp64| {float(c['ht'].abs().max()):.2e} |ref-fp64| {float(c['rt'].abs().max()):.2e}   "
              f"p99.9 |hip-ref| {float(c['hr'].abs().quantile(0.999)):.2e}")
