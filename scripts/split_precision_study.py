"""Error of the rendered 256-d features (Sum_s w_s h3_s along each ray) when the NeRF MLP's matrix products are evaluated with
operand splits of different widths -- CPU emulation (fp64 accumulate of exactly rounded operands) on a synthetic scene.

  bf16x3   w_hi x_hi + w_hi x_lo + w_lo x_hi, bf16 parts      (the shipped kernel's arithmetic)
  fp16x3   the same with fp16 parts
  fp16x2w  (w_hi + w_lo) x_hi: activations rounded to fp16, weights kept to 22 bits           (2 MFMAs per product block)
  fp16x2x  w_hi (x_hi + x_lo): weights rounded to fp16                                         (2 MFMAs)
  fp16x1 / bf16x1  both rounded once                                                            (1 MFMA)
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import synth
from oracle import nerf_oracle as no  # (analysis script, not product code)

torch.manual_seed(0)
S, R = 64, 256
sd = {k: v.double() for k, v in synth.nerf_state_dict(seed=0, density_bias=3.0).items()}
H, W = 480, 640
K, pose = synth.intrinsics(H, W), torch.as_tensor(synth.camera_pose(1), dtype=torch.float32)
from oracle import inerf_oracle as io
rays = io.gen_rays(pose, W, H, torch.as_tensor(K, dtype=torch.float32))[1000:1000 + R].detach()
t = no.sample_coarse(rays, S, torch.rand(R, S + 1))
mean, var = no.frustum_gaussians(t, rays[:, :3], rays[:, 3:6], rays[:, 11:12])
x_pts = no.ipe(mean.reshape(-1, 3), var.reshape(-1, 3), 15).double()


def rnd(x, dt):
    return x.to(torch.float32).to(dt).to(torch.float64)


def split(x, dt):
    hi = rnd(x, dt)
    return hi, rnd(x - hi, dt)


def product(x, w, scheme):
    if scheme == "fp64":
        return x @ w.T
    dt = torch.bfloat16 if scheme.startswith("bf16") else torch.float16
    xh, xl = split(x, dt)
    wh, wl = split(w, dt)
    kind = scheme[4:]
    if kind == "x3":
        return xh @ wh.T + xl @ wh.T + xh @ wl.T
    if kind == "x2w":
        return xh @ (wh + wl).T
    if kind == "x2x":
        return (xh + xl) @ wh.T
    return xh @ wh.T


def features(scheme):
    h = x_pts
    tap = None
    for i in range(8):
        w, b = sd[f"nerf_fine.pts_linears.{i}.weight"], sd[f"nerf_fine.pts_linears.{i}.bias"]
        h = torch.relu(product(h, w, scheme).float().double() + b)  # fp32 accumulator, as the MFMA returns it
        if i == 3:
            tap = h
        if i == 4:
            h = torch.cat([x_pts, h], -1)
    sigma = product(h, sd["nerf_fine.alpha_linear.weight"], scheme) + sd["nerf_fine.alpha_linear.bias"]
    raw = torch.cat([torch.zeros(R * S, 3, dtype=torch.float64), sigma], -1).reshape(R, S, 4)
    wts = no.composite(raw.float(), t, rays[:, 3:6])[3].double()
    return (wts[..., None] * tap.reshape(R, S, -1)).sum(1)


ref = features("fp64")
print(f"{R} rays x {S} samples; feature magnitude max {ref.abs().max():.2f}")
for scheme in ("bf16x3", "fp16x3", "fp16x2w", "fp16x2x", "fp16x1", "bf16x1"):
    f = features(scheme)
    print(f"  {scheme:8s} max |feature error| {float((f - ref).abs().max()):.2e}   rms {float((f - ref).pow(2).mean().sqrt()):.2e}")


# ---- a cheaper split for the COARSE pass only (its weights feed nothing but the resampler): error of the FINAL features
def sigma_of(prefix, xin, scheme):
    h = xin
    for i in range(8):
        w, b = sd[f"{prefix}.pts_linears.{i}.weight"], sd[f"{prefix}.pts_linears.{i}.bias"]
        h = torch.relu(product(h, w, scheme).float().double() + b)
        if i == 3:
            tap = h
        if i == 4:
            h = torch.cat([xin, h], -1)
    return product(h, sd[f"{prefix}.alpha_linear.weight"], scheme) + sd[f"{prefix}.alpha_linear.bias"], tap


def render(coarse_scheme, fine_scheme, jitter):
    sg, _ = sigma_of("nerf_coarse", x_pts, coarse_scheme)
    raw = torch.cat([torch.zeros(R * S, 3, dtype=torch.float64), sg], -1).reshape(R, S, 4)
    w_c = no.composite(raw.float(), t, rays[:, 3:6])[3]
    t_f = no.resample(t, w_c, jitter, padding=0.01, randomized=True)
    m2, v2 = no.frustum_gaussians(t_f, rays[:, :3], rays[:, 3:6], rays[:, 11:12])
    xf = no.ipe(m2.reshape(-1, 3), v2.reshape(-1, 3), 15).double()
    sg, tap = sigma_of("nerf_fine", xf, fine_scheme)
    raw = torch.cat([torch.zeros(R * S, 3, dtype=torch.float64), sg], -1).reshape(R, S, 4)
    wts = no.composite(raw.float(), t_f, rays[:, 3:6])[3].double()
    return (wts[..., None] * tap.reshape(R, S, -1)).sum(1), t_f


jit = torch.rand(R, S + 1) * (1.0 / (S + 1) - 1.2e-7)
ref, t_ref = render("fp64", "fp64", jit)
print("coarse pass in a cheaper split, fine pass bf16x3 -- error of the final features / of the fine fence posts:")
for cs in ("bf16x3", "fp16x2w", "fp16x1", "bf16x2w", "bf16x1"):
    f, tf = render(cs, "bf16x3", jit)
    print(f"  coarse {cs:8s} max |feature error| {float((f - ref).abs().max()):.2e}   max |t_fine error| {float((tf - t_ref).abs().max()):.2e}")


# ---- mixed arithmetic INSIDE the fine pass: tapped layers 0..3 on bf16x3, the density branch (layers 4..7 + alpha head) cheaper
def sigma_mixed(xin, lo_scheme, hi_scheme):
    h = xin
    for i in range(8):
        w, b = sd[f"nerf_fine.pts_linears.{i}.weight"], sd[f"nerf_fine.pts_linears.{i}.bias"]
        h = torch.relu(product(h, w, lo_scheme if i <= 3 else hi_scheme).float().double() + b)
        if i == 3:
            tap = h
        if i == 4:
            h = torch.cat([xin, h], -1)
    return product(h, sd["nerf_fine.alpha_linear.weight"], hi_scheme) + sd["nerf_fine.alpha_linear.bias"], tap


def render_mixed(hi_scheme):
    sg, _ = sigma_of("nerf_coarse", x_pts, "fp16x1")
    raw = torch.cat([torch.zeros(R * S, 3, dtype=torch.float64), sg], -1).reshape(R, S, 4)
    w_c = no.composite(raw.float(), t, rays[:, 3:6])[3]
    t_f = no.resample(t, w_c, jit, padding=0.01, randomized=True)
    m2, v2 = no.frustum_gaussians(t_f, rays[:, :3], rays[:, 3:6], rays[:, 11:12])
    xf = no.ipe(m2.reshape(-1, 3), v2.reshape(-1, 3), 15).double()
    sg, tap = sigma_mixed(xf, "bf16x3", hi_scheme)
    raw = torch.cat([torch.zeros(R * S, 3, dtype=torch.float64), sg], -1).reshape(R, S, 4)
    wts = no.composite(raw.float(), t_f, rays[:, 3:6])[3].double()
    return (wts[..., None] * tap.reshape(R, S, -1)).sum(1), (wts[..., None] * m2.double()).sum(1)


f_ref, p_ref = render_mixed("fp64") if False else (None, None)
sg_ref, tap_ref = sigma_mixed(no.ipe(*[x.reshape(-1, 3) for x in no.frustum_gaussians(t_ref, rays[:, :3], rays[:, 3:6], rays[:, 11:12])], 15).double(), "fp64", "fp64")
raw = torch.cat([torch.zeros(R * S, 3, dtype=torch.float64), sg_ref], -1).reshape(R, S, 4)
w_ref = no.composite(raw.float(), t_ref, rays[:, 3:6])[3].double()
m_ref = no.frustum_gaussians(t_ref, rays[:, :3], rays[:, 3:6], rays[:, 11:12])[0].double()
f_ref, p_ref = (w_ref[..., None] * tap_ref.reshape(R, S, -1)).sum(1), (w_ref[..., None] * m_ref).sum(1)
print("fine pass: layers 0-3 bf16x3, density branch (layers 4-7 + alpha) in a cheaper split -- error of features / of the rendered points:")
for hs in ("bf16x3", "fp16x2w", "fp16x1"):
    f, pnt = render_mixed(hs)
    print(f"  density branch {hs:8s} max |feature error| {float((f - f_ref).abs().max()):.2e}   max |point error| {float((pnt - p_ref).abs().max()):.2e}")
