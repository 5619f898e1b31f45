import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import load_golden
from nerfmatch_amd import ops
from test_nerf_gpu import make_renderer
from test_surface_seeds_gpu import SEEDS
from test_resample_truth_gpu import resample_fp64
gpu = torch.device("cuda:0")
for precision in ("fp16x3", "fp32", "bf16x3"):
    H, G = [], []
    for ws, ps in SEEDS:
        fx = load_golden(f"nerf_surf_w{ws}_p{ps}")
        ren, sd = make_renderer(fx, gpu)
        rays, t_c, jit = fx["rays"].to(gpu), fx["t_coarse"], fx["jitter"]
        w64 = fx["comp_weights"].double() + fx["truth_d_weights_coarse"].double()
        truth = resample_fp64(t_c, w64, jit)
        w_hip = ren.nerf_coarse.fused(precision, rays, t_c.to(gpu), None, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
        hip = ops.resample(t_c.to(gpu), w_hip, jit.to(gpu)).cpu().double()
        H.append((hip - truth).abs().max(-1)[0]); G.append((fx["t_fine"].double() - truth).abs().max(-1)[0])
        if precision == "fp16x3":
            ew = (w_hip.cpu().double() - w64).abs(); eg = fx["truth_d_weights_coarse"].double().abs()
            print(f"  w{ws} p{ps} coarse weights |hip-fp64| max {float(ew.max()):.2e} rms {float(ew.pow(2).mean().sqrt()):.2e}   reference max {float(eg.max()):.2e} rms {float(eg.pow(2).mean().sqrt()):.2e}")
    H, G = torch.cat(H), torch.cat(G)
    qs = torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64)
    print(precision, "hip q", [f"{float(x):.2e}" for x in torch.quantile(H, qs)], "max", f"{float(H.max()):.2e}", "n>1e-5", int((H > 1e-5).sum()), "n>1e-4", int((H > 1e-4).sum()))
    print(precision, "ref q", [f"{float(x):.2e}" for x in torch.quantile(G, qs)], "max", f"{float(G.max()):.2e}", "n>1e-5", int((G > 1e-5).sum()), "n>1e-4", int((G > 1e-4).sum()))
