"""SURVEY 8a row N0, direct: the HIP encodings against the reference's own values (VERDICT r4 "What's weak" 1).

The fixtures' `ipe_coarse` / `dir_pe` are the outputs of the reference's `renderer.xyz_encoder(mean, y=var)[0]` and
`renderer.dirs_encoder(viewdirs)` (tests/golden/make_golden.py:156-158).  Compared here, at 2e-7 absolute (values are in [-1, 1]):
  * the callable modules (nerfmatch_amd/nerf/embedding.py -> nm_mip_encode / nm_fourier_embed), which the reference's evaluator
    reaches into (nerfmatch_evaluator.py:385-393);
  * nm_inerf_encode (the encode step of the iNeRF fine pass);
  * arith = 1: the exp2 / fp32-sine device functions the split render kernels inline (their encoding pinned by value, not only
    through the MLP's outputs) -- stated bound 3e-7, measured rms beside the correctly rounded arithmetic's;
and, for the fixtures that store no per-sample encodings, against the oracle (itself pinned to the reference at 0 ulp
by tests/test_oracle_golden.py) on the frustum Gaussians of their coarse fence posts.  A 2^14 x argument case with |x| up to 1 is included
(arguments up to 16384 rad) with an fp64 truth beside it."""
import ctypes as C

import pytest
import torch

from conftest import load_golden
from nerfmatch_amd import _lib, ops
from nerfmatch_amd.nerf.embedding import FourierEmbedding, PositionalEncodingMIP
from oracle import matcher_oracle as mo
from oracle import nerf_oracle as no

pytestmark = pytest.mark.gpu

WITH_IPE = ["nerf_r32_s32", "nerf_r32_s32_last", "nerf_r128_s64_app", "nerf_surface_r256_s64_app", "nerf_surface_r512_s128"]
SURF = [f"nerf_surf_w{w}_p{p}" for w, p in zip((1, 1, 2, 2, 3, 3, 4, 4, 5, 5), range(21, 31))]
ALL_NERF = WITH_IPE + SURF + ["nerf_fine_count_c32_f64"]  # every NeRF fixture with fence posts (nerf_far_fallback holds rays only)
TOL = 2e-7


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def ipe_truth(x, y, F=15):
    """fp64 evaluation of the reference's expression on the fp32 arguments it forms (x * 2^s exact, `+ 0.5 * pi` an fp32 sum)."""
    x, y = x.cpu().float(), y.cpu().float()
    xe = (x[:, None, :] * (2 ** torch.arange(F))[:, None]).reshape(x.shape[0], -1)
    xe = torch.cat((xe, xe + 0.5 * torch.pi), -1).double()
    ye = (y.double()[:, None, :] * (4.0 ** torch.arange(F))[:, None]).reshape(x.shape[0], -1).repeat(1, 2)
    return torch.exp(-0.5 * ye) * torch.sin(xe)


@pytest.mark.parametrize("name", WITH_IPE)
def test_modules_vs_reference_values(gpu, built_lib, name):
    fx = load_golden(name)
    n = fx["ipe_coarse"].shape[0]
    mean, var = fx["mean_coarse"].reshape(-1, 3)[:n].to(gpu), fx["var_coarse"].reshape(-1, 3)[:n].to(gpu)
    enc = PositionalEncodingMIP(15).to(gpu)
    x_ret, y_ret = enc(mean, var)
    assert x_ret.shape == (n, 90) and y_ret.shape == (n, 90)
    assert maxdiff(x_ret, fx["ipe_coarse"]) <= TOL
    # leading dimensions are kept like the reference's reshape
    x3, _ = enc(mean.reshape(-1, fx["S"], 3), var.reshape(-1, fx["S"], 3))
    assert x3.shape == (n // fx["S"], fx["S"], 90) and torch.equal(x3.reshape(n, 90), x_ret)
    dpe = PositionalEncodingMIP(4).to(gpu)(fx["rays"][:, 8:11].to(gpu))
    assert dpe.shape == fx["dir_pe"].shape and maxdiff(dpe, fx["dir_pe"]) <= TOL
    # the second return value (variance of the encoding; embedding.py:75-78) against the formula in fp64
    xe = (mean.cpu().double()[:, None, :] * (2.0 ** torch.arange(15))[:, None]).reshape(n, -1)
    xe = torch.cat((xe, (xe.float() + 0.5 * torch.pi).double()), -1)
    ye = (var.cpu().double()[:, None, :] * (4.0 ** torch.arange(15))[:, None]).reshape(n, -1).repeat(1, 2)
    xr = torch.exp(-0.5 * ye) * torch.sin(xe)
    y_truth = torch.clamp(0.5 * (1 - torch.exp(-2 * ye) * torch.cos(2 * xe)) - xr**2, min=0)
    assert maxdiff(y_ret, y_truth) <= 4e-7  # (a difference of O(1) terms: two roundings)


@pytest.mark.parametrize("name", WITH_IPE)
def test_split_kernel_arithmetic_vs_reference_values(gpu, built_lib, name):
    """arith = 1 evaluates x_ret with the device functions nerf_fwd_bf16.hip inlines (exp2-based exponential, fp32 Cody-Waite sine)."""
    fx = load_golden(name)
    n = fx["ipe_coarse"].shape[0]
    mean, var = fx["mean_coarse"].reshape(-1, 3)[:n].to(gpu), fx["var_coarse"].reshape(-1, 3)[:n].to(gpu)
    enc = PositionalEncodingMIP(15).to(gpu)
    exact = enc(mean, var)[0]
    enc.arith = 1
    fast = enc(mean, var)[0]
    truth = ipe_truth(mean, var)
    ref = fx["ipe_coarse"].double()
    rms = lambda a: float((a.cpu().double() - truth).pow(2).mean().sqrt())
    print(f"{name}: rms error against fp64  reference fp32 {rms(ref):.2e}  arith 0 {rms(exact):.2e}  arith 1 (split kernels) {rms(fast):.2e}; "
          f"max |arith 1 - reference| {maxdiff(fast, ref):.2e}")
    assert maxdiff(fast, ref) <= 3e-7
    assert rms(fast) <= 2.0 * max(rms(ref), 1e-8)


@pytest.mark.parametrize("name", ALL_NERF)
def test_modules_and_inerf_encode_vs_oracle_on_every_fixture(gpu, built_lib, name):
    fx = load_golden(name)
    rays = fx["rays"]
    t = fx["t_coarse"] if "t_coarse" in fx else no.sample_coarse(rays, fx["S_coarse"], fx["t_rand"])
    R, S = rays.shape[0], t.shape[1] - 1
    mean, var = no.frustum_gaussians(t, rays[:, 0:3], rays[:, 3:6], rays[:, 11:12])
    mean, var = mean.reshape(-1, 3), var.reshape(-1, 3)
    want = no.ipe(mean, var, 15)
    if "ipe_coarse" in fx:  # the oracle's values are the reference's there (bit for bit on the host that made the fixtures; another
        assert maxdiff(want[: fx["ipe_coarse"].shape[0]], fx["ipe_coarse"]) <= 1.2e-7  # CPU's vector sine may differ in the last bit)
    got = PositionalEncodingMIP(15).to(gpu)(mean.to(gpu), var.to(gpu))[0]
    assert maxdiff(got, want) <= TOL
    want_d = no.dir_pe(rays[:, 8:11], 4)
    assert maxdiff(PositionalEncodingMIP(4).to(gpu)(rays[:, 8:11].to(gpu)), want_d) <= TOL
    # nm_inerf_encode: frustum Gaussian + IPE + direction PE of the iNeRF fine pass, from rays and fence posts
    n = R * S
    xi, xd = torch.empty(n, 96, device=gpu), torch.empty(n, 48, device=gpu)
    rays_d, t_d = rays.to(gpu).contiguous(), t.to(gpu).contiguous()  # (named: a temporary would be freed before the kernel runs)
    _lib.check(_lib.lib().nm_inerf_encode(_lib.dptr(rays_d), _lib.dptr(t_d), R, S, S, None, _lib.dptr(xi), _lib.dptr(xd), ops.stream()),
               "nm_inerf_encode")
    # its Gaussians are its own fp32 evaluation of the frustum formulas (the iNeRF loop re-implements the sampling inline,
    # nerfmatch_evaluator.py:364-383: hw = (t0 - t1) / 2 -- the sign flip cancels in hw^2 -- and scale_var = 1): an encoding of
    # arguments up to 2^14 x sees a 1-ulp difference of the mean as up to 2^14 * 6e-8 = 1e-3 rad, so the comparison is made where the
    # inputs agree bit for bit, and the rest is bounded through the damping the reference applies itself (exp(-var 4^s / 2) <= 1)
    assert maxdiff(xd[:, :27], want_d[:, None, :].expand(R, S, 27).reshape(n, 27)) <= TOL
    assert float(xd[:, 27:].abs().max()) == 0.0 and float(xi[:, 90:].abs().max()) == 0.0
    # scales 2^0 .. 2^3 (both halves below): 1 ulp of the mean moves the argument by < 1e-6
    cols = list(range(12)) + list(range(45, 57))
    assert maxdiff(xi[:, cols], want[:, cols]) <= 2e-6
    assert maxdiff(xi[:, :90], want) <= 2e-3


def test_large_argument_case(gpu, built_lib):
    """|x| up to 1 at scale 2^14: arguments up to 16384 rad, tiny variance (damping ~1) -- the regime where a sloppy range reduction
    shows.  Against fp64 truth: the correctly rounded class (arith 0) within 1e-7 (like torch's own fp32 evaluation), the split kernels' sine
    (arith 1) within its stated 1e-7 (+ the damping's rounding)."""
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(4096, 3, generator=g) * 2 - 1)
    y = torch.full_like(x, 1e-12)
    enc = PositionalEncodingMIP(15).to(gpu)
    truth = ipe_truth(x, y)
    ref32 = no.ipe(x, y, 15)
    e0 = maxdiff(enc(x.to(gpu), y.to(gpu))[0], truth)
    enc.arith = 1
    e1 = maxdiff(enc(x.to(gpu), y.to(gpu))[0], truth)
    print(f"2^14 x case: max error against fp64  torch-CPU fp32 {maxdiff(ref32, truth):.2e}  arith 0 {e0:.2e}  arith 1 {e1:.2e}")
    assert e0 <= 1.0e-7 and e1 <= 2e-7  # (two roundings -- damping, product -- of values up to 1: the reference's own fp32 run is at 8.4e-8)
    assert maxdiff(enc(x.to(gpu), y.to(gpu))[0], ref32) <= 3e-7


def test_fourier_embedding_module(gpu, built_lib):
    fx = load_golden("matcher_c2f")
    pt3d = fx["pt3d"][0]
    emb = FourierEmbedding(15).to(gpu)
    out = emb(pt3d.to(gpu))
    assert out.shape == (pt3d.shape[0], 93)
    assert maxdiff(out, fx["fourier_pt3d"][0]) < 1e-6  # un-normalised metres at 2^14: the reference's fp32 sine itself is ~5e-7 from fp64 there
    assert maxdiff(out, mo.fourier_embed(pt3d.double()).float()) <= 6.5e-8  # fp64 truth, rounded once
    big = torch.tensor([[37.25, -81.5, 12.125], [0.0, 1e-3, -250.0]])
    assert maxdiff(emb(big.to(gpu)), mo.fourier_embed(big.double()).float()) <= 6.5e-8
    lead = emb(pt3d.reshape(2, -1, 3).to(gpu))
    assert lead.shape == (2, pt3d.shape[0] // 2, 93) and torch.equal(lead.reshape(-1, 93), out)
    with pytest.raises(NotImplementedError):
        FourierEmbedding(15, logscale=False)


def test_renderer_encoders_are_callable(gpu, built_lib):
    """The attribute path the reference's evaluator uses: renderer.xyz_encoder(mean, var)[0], renderer.dirs_encoder(viewdirs)."""
    from nerfmatch_amd import synth
    from nerfmatch_amd.nerf.renderer import NerfRenderer

    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=32), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0), strict=True)
    ren.to(gpu).eval()
    fx = load_golden("nerf_r32_s32")
    n = fx["ipe_coarse"].shape[0]
    x = ren.xyz_encoder(fx["mean_coarse"].reshape(-1, 3)[:n].to(gpu), fx["var_coarse"].reshape(-1, 3)[:n].to(gpu))[0]
    d = ren.dirs_encoder(fx["rays"][:, 8:11].to(gpu))
    assert maxdiff(x, fx["ipe_coarse"]) <= TOL and maxdiff(d, fx["dir_pe"]) <= TOL
    assert ren.state_dict()["xyz_encoder.scales"].dtype == torch.int64
