"""Round 4, before touching the kernel: WHERE does the fp16 hi/lo split lose accuracy on a trained-like scene, and what buys it back?

CPU emulation of the fused pass's layer products on the surface fixture's inputs (exactly rounded operands, fp64 product sums,
ONE fp32 rounding of the accumulator per MFMA = per 16-wide K-step and product, like v_mfma_f32_32x32x16_f16), against
  * the fp64 evaluation of the same network on the same fp32 inputs ("truth"), and
  * the oracle (torch-CPU fp32), which is what the parity tests compare with.

Variants of the split (weights are always split on the host with round-to-nearest hi and lo):
  shipped   activations: hi by round-toward-zero (v_cvt_pkrtz), lo rounded to nearest; no scaling (lo parts of anything below
            2^-3 are fp16 subnormals: absolute quantum 2^-24)
  rne       activations: hi rounded to nearest (v_cvt_pk_f16_f32) -- halves |lo| and with it the dropped lo*lo term
  scaled    power-of-two scaling of weights (2^a: max|W| -> [2^13, 2^14)) and activations (2^c: max|x| -> [2^10, 2^11)): no subnormal parts
  rne+scaled
  x4        all four products (adds w_lo * x_lo): the ceiling of any three-product variant
  fp32mfma  v_mfma_f32_32x32x2_f32: exact products, fp32 accumulate, one rounding per 2-wide K-step (the fp32 kernel)

    python scripts/fp16x3_scaling_study.py [seed ...]
"""
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from nerfmatch_amd import synth
from oracle import nerf_oracle as no  # (analysis script, not product code)

torch.set_grad_enabled(False)
H, W, S, R_USE = 128, 256, 64, 160


def rtz_f16(x):
    """fp64 tensor of fp32 values -> truncated to fp16 precision (11 significant bits; quantum 2^-24 below 2^-14), saturating"""
    x = x.clamp(-65504.0, 65504.0)
    m, e = torch.frexp(x)  # x = m * 2^e, 0.5 <= |m| < 1
    q = torch.where(e > -13, torch.ldexp(torch.ones_like(x), e - 11), torch.full_like(x, 2.0**-24))
    return torch.trunc(x / q) * q


def rne_f16(x):
    return x.clamp(-65504.0, 65504.0).to(torch.float32).to(torch.float16).to(torch.float64)


def split_x(x, mode):
    hi = rtz_f16(x) if mode == "rtz" else rne_f16(x)
    return hi, rne_f16(x - hi)


def pow2_floor(v):
    return 2.0 ** math.floor(math.log2(v))


class Emu:
    def __init__(self, xmode="rtz", scaled=False, four=False, fp32mfma=False):
        self.xmode, self.scaled, self.four, self.fp32mfma = xmode, scaled, four, fp32mfma

    def product(self, x, w):
        """x (n,K) fp64 holding fp32 values, w (N,K) -> fp32-accumulated product as fp64 tensor"""
        n, K = x.shape
        if self.fp32mfma:
            acc = torch.zeros(n, w.shape[0], dtype=torch.float32)
            for k in range(0, K, 2):
                acc = (acc.double() + x[:, k : k + 2] @ w[:, k : k + 2].T).float()
            return acc.double()
        sw = sx = 1.0
        if self.scaled:
            sw = 2.0**13 / pow2_floor(float(w.abs().max()))
            sx = 2.0**10 / pow2_floor(max(float(x.abs().max()), 1e-30))
        wh = rne_f16(w * sw)
        wl = rne_f16(w * sw - wh)
        xh, xl = split_x(x * sx, self.xmode)
        acc = torch.zeros(n, w.shape[0], dtype=torch.float32)
        for k in range(0, K, 16):
            sl = slice(k, k + 16)
            acc = (acc.double() + xh[:, sl] @ wh[:, sl].T).float()
            acc = (acc.double() + xl[:, sl] @ wh[:, sl].T).float()
            acc = (acc.double() + xh[:, sl] @ wl[:, sl].T).float()
            if self.four:
                acc = (acc.double() + xl[:, sl] @ wl[:, sl].T).float()
        return acc.double() / (sw * sx)


def mlp(sd, prefix, x_pts, prod, f32_epilogue=True):
    """layers 0..7 + density head; returns (sigma_raw, h3, h7).  prod(x, w) -> pre-activation without bias."""
    rd = (lambda v: v.float().double()) if f32_epilogue else (lambda v: v)
    h = x_pts
    taps = {}
    for i in range(8):
        w, b = sd[f"{prefix}.pts_linears.{i}.weight"].double(), sd[f"{prefix}.pts_linears.{i}.bias"].double()
        if i == 5:  # kernel order: hidden K-steps, then the skip connection's IPE K-steps, one accumulator
            h = torch.cat([h, x_pts], -1)
            w = torch.cat([w[:, 90:], w[:, :90]], -1)
        h = torch.relu(rd(prod(h, w) + b))
        taps[i] = h
    wa, ba = sd[f"{prefix}.alpha_linear.weight"].double(), sd[f"{prefix}.alpha_linear.bias"].double()
    sigma = rd(h @ wa.T + ba)  # (the kernel's density head is a plain fp32 FMA chain; its rounding is not the subject here)
    return sigma[:, 0], taps[3], taps[7]


def composite64(sigma, t, d, feats):
    sg = torch.relu(sigma)
    delta = (t[:, 1:] - t[:, :-1]).double() * d.double().norm(dim=-1, keepdim=True)
    alpha = 1.0 - torch.exp(-sg * delta)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * trans
    return w, (w[..., None] * feats).sum(1)


def study(seed, pose_seed):
    sd = synth.nerf_state_dict(seed=seed, style="surface")
    K = torch.tensor([[240.0, 0, W / 2], [0, 240.0, H / 2], [0, 0, 1]])
    rays = no.make_rays(H, W, K, synth.camera_pose(pose_seed))[:R_USE]
    R = rays.shape[0]
    t = no.sample_coarse(rays, S, synth.uniform01((R, S + 1), 1000 + seed))
    mean, var = no.frustum_gaussians(t, rays[:, :3], rays[:, 3:6], rays[:, 11:12])
    x32 = no.ipe(mean.reshape(-1, 3), var.reshape(-1, 3), 15)
    x = x32.double()
    d = rays[:, 3:6]

    def run(prod, f32_epilogue=True):
        sg, h3, h7 = mlp(sd, "nerf_coarse", x, prod, f32_epilogue)
        w, f7 = composite64(sg.reshape(R, S), t, d, h7.reshape(R, S, -1))
        _, f3 = composite64(sg.reshape(R, S), t, d, h3.reshape(R, S, -1))
        return dict(sigma=sg, h3=h3, h7=h7, w=w, f7=f7, f3=f3)

    truth = run(lambda a, b: a @ b.T, f32_epilogue=False)
    # the oracle proper (torch fp32 end to end, its own compositing)
    raw, feats = no.nerf_mlp(sd, "nerf_coarse", x32, torch.zeros(R * S, 27), None, stop_layer=-1)
    _, _, _, w_or = no.composite(raw.reshape(R, S, 4), t, d)
    orc = dict(sigma=raw[:, 3].double(), h7=feats.double(), w=w_or.double(), f7=(w_or[..., None] * feats.reshape(R, S, -1)).sum(1).double())
    print(f"seed {seed} pose {pose_seed}: {R} rays x {S} samples; |h7| max {float(truth['h7'].abs().max()):.1f}  |h3| max {float(truth['h3'].abs().max()):.1f}  "
          f"sigma {float(truth['sigma'].min()):.0f}..{float(truth['sigma'].max()):.0f}  feat(last) max {float(truth['f7'].abs().max()):.1f}  median max-weight {float(truth['w'].max(-1)[0].median()):.2f}")

    def line(name, r):
        e = lambda k, ref: float((r[k] - ref[k]).abs().max())
        print(f"  {name:11s} vs truth: sigma {e('sigma', truth):.2e}  h7 {e('h7', truth):.2e}  w {e('w', truth):.2e}  feat {e('f7', truth):.2e}   "
              f"| vs oracle: sigma {e('sigma', orc):.2e}  w {e('w', orc):.2e}  feat {e('f7', orc):.2e}")

    line("oracle", orc)
    out = {}
    for name, emu in (("fp32mfma", Emu(fp32mfma=True)), ("shipped", Emu("rtz")), ("rne", Emu("rne")), ("scaled", Emu("rtz", scaled=True)),
                      ("rne+scaled", Emu("rne", scaled=True)), ("x4", Emu("rne", scaled=True, four=True))):
        r = run(emu.product)
        line(name, r)
        out[name] = float((r["f7"] - orc["f7"]).abs().max())
    return out


if __name__ == "__main__":
    seeds = [int(a) for a in sys.argv[1:]] or [0]
    for sdx in seeds:
        study(sdx, 11 + sdx)
