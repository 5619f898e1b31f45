"""Which rays of the trained-like end-to-end render / resampling chain are beyond 1e-4, and is each explained by a moved fence post?
(study behind tests/test_surface_seeds_gpu.py::test_surface_seed_render_end_to_end and tests/test_resample_truth_gpu.py)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from conftest import load_golden
from nerfmatch_amd import ops
from test_nerf_gpu import make_renderer
from test_resample_truth_gpu import resample_fp64
from test_surface_seeds_gpu import SEEDS

torch.set_grad_enabled(False)
gpu = torch.device("cuda:0")


def bins_of(t_new, t_c):
    """index of the coarse interval [t_c[j], t_c[j+1]) every new fence post lies in"""
    return (torch.searchsorted(t_c.double().contiguous(), t_new.double().contiguous(), right=True) - 1).clamp(0, t_c.shape[1] - 2)


for precision in ("fp16x3", "fp32"):
    print("=====", precision)
    for ws, ps in SEEDS:
        fx = load_golden(f"nerf_surf_w{ws}_p{ps}")
        ren, sd = make_renderer(fx, gpu)
        ren.precision, ren.ret_pfeat = precision, True
        preds = ren.predict(fx["rays"].to(gpu), 1, 1, out_raw=True, t_rand=fx["t_rand"], jitter=fx["jitter"], debug=True)
        t_h, t_r = preds["t_fine"].cpu(), fx["t_fine"]
        dt = (t_h - t_r).abs().max(-1)[0]
        moved_bin = (bins_of(t_h, fx["t_coarse"]) != bins_of(t_r, fx["t_coarse"])).any(-1)
        for k in ("feat_fine", "pts_fine", "rgb_fine", "depth_fine"):
            d = (preds[k].cpu() - fx[f"pred_{k}"]).abs().reshape(t_h.shape[0], -1).max(-1)[0]
            big = torch.nonzero(d > 1e-4).flatten().tolist()
            still = dt <= 2e-6
            print(f"w{ws} p{ps} {k}: rays>1e-4 {len(big)}: " + " ".join(f"[ray {r} err {float(d[r]):.1e} dt {float(dt[r]):.1e} bin_moved {bool(moved_bin[r])}]" for r in big) +
                  f" | rays with dt<=2e-6: {int(still.sum())}, their max err {float(d[still].max()) if still.any() else 0:.1e}; dt<=1e-5: max err {float(d[dt <= 1e-5].max()):.1e}")
        # chain: hip coarse weights -> hip resampler against the fp64 fence posts
        w64 = fx["comp_weights"].double() + fx["truth_d_weights_coarse"].double()
        truth = resample_fp64(fx["t_coarse"], w64, fx["jitter"])
        e_h = (t_h.double() - truth).abs().max(-1)[0]
        e_r = (t_r.double() - truth).abs().max(-1)[0]
        bh, br, bt = bins_of(t_h, fx["t_coarse"]), bins_of(t_r, fx["t_coarse"]), bins_of(truth, fx["t_coarse"])
        mh, mr = (bh != bt).any(-1), (br != bt).any(-1)
        for nm, e, m in (("hip", e_h, mh), ("ref", e_r, mr)):
            big = torch.nonzero(e > 1e-4).flatten().tolist()
            print(f"   chain {nm}: rays>1e-4 {len(big)} " + " ".join(f"[ray {r} err {float(e[r]):.1e} bin_moved {bool(m[r])}]" for r in big) +
                  f" | same-bin rays {int((~m).sum())}: max err {float(e[~m].max()):.1e}; moved-bin rays {int(m.sum())}: max {float(e[m].max()) if m.any() else 0:.1e}")
