"""Path sweep of the split NeRF kernels (round 4): every tap layer x colour heads on / off x row length x zero-tail skip x feature
combination, against the fp32 kernel on the same inputs.

The fp32 kernel (csrc/nerf_fwd.hip) is an independent implementation of the same pass: it runs feature_linear as a layer (the split
kernels fold it into the views layer at pack time), keeps the tapped activations in registers (the split kernels park them in a workspace
and bring them back by LDS-DMA behind the last K-loop, from a different point of the tile with and without colour heads) and reduces them
with its own code (the split kernels use the lane reduce-scatter).  It is itself pinned to the reference's golden fixtures in
test_nerf_gpu.py; here it is the oracle for the control-flow variants those fixtures do not reach.
"""
import pytest
import torch

from nerfmatch_amd import ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer

pytestmark = pytest.mark.gpu


def _scene(gpu, S, R, seed):
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=S), training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=seed, density_bias=3.0))
    ren.to(gpu).eval()
    rays = ops.raygen(synth.intrinsics(), synth.camera_pose(seed), 480, 640, gpu)[0][:R].contiguous()
    g = torch.Generator(device="cpu").manual_seed(seed)
    t_c = ops.sample_coarse(rays, torch.rand(R, S + 1, generator=g).to(gpu), S)
    return ren, rays, t_c


def _close(a, b, tol, what):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err < tol, f"{what}: {err:.2e} of scale {scale:.3g}"


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
@pytest.mark.parametrize("tap", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("need_rgb", [True, False])
def test_every_tap_layer_with_and_without_colour_heads(gpu, built_lib, precision, tap, need_rgb):
    S, R = 64, 300  # 150 tiles of two rays: more tiles than CUs' worth of one wave, ragged last workgroups
    ren, rays, t = _scene(gpu, S, R, seed=tap + 1)
    with torch.no_grad():
        ref = ops.nerf_fwd(ren.nerf_fine.packed(gpu, "fp32"), rays, t, tap_layer=tap, need_rgb=need_rgb)
        out = ops.nerf_fwd(ren.nerf_fine.packed(gpu, precision), rays, t, tap_layer=tap, need_rgb=need_rgb)
    tol = 2e-5 if precision == "fp16x3" else 2e-3  # bf16 split: 16 significant bits
    for k in ("weights", "feat", "pts", "depth", "acc") + (("rgb",) if need_rgb else ()):
        _close(out[k], ref[k], tol, f"{k} tap {tap} rgb {need_rgb}")


@pytest.mark.parametrize("S,R", [(32, 517), (64, 1), (64, 129), (128, 77), (256, 41), (384, 9)])
@pytest.mark.parametrize("feat_max", [False, True])
def test_row_lengths_and_feature_combination(gpu, built_lib, S, R, feat_max):
    """S = 32: four rays per tile; 128: one; 256 / 384: several chunks per ray (the read-back and the reduction run once per chunk,
    the running feature lives across them)."""
    ren, rays, t = _scene(gpu, S, R, seed=S + R)
    with torch.no_grad():
        ref = ops.nerf_fwd(ren.nerf_fine.packed(gpu, "fp32"), rays, t, tap_layer=3, feat_max=feat_max)
        out = ops.nerf_fwd(ren.nerf_fine.packed(gpu, "fp16x3"), rays, t, tap_layer=3, feat_max=feat_max)
    for k in ("weights", "pts", "rgb", "depth", "acc"):
        _close(out[k], ref[k], 2e-5, f"{k} S {S}")
    if feat_max:
        # the selected sample is an argmax over weights that agree to ~1e-6: rays whose two best weights are closer than that may pick
        # another sample in the two arithmetics -- compare the rays with a clear winner
        w = ref["weights"]
        top2 = w.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert int(clear.sum()) >= R // 2 or R < 4
        _close(out["feat"][clear], ref["feat"][clear], 2e-5, f"feat(max) S {S}")
    else:
        _close(out["feat"], ref["feat"], 2e-5, f"feat S {S}")


@pytest.mark.parametrize("need_rgb", [True, False])
@pytest.mark.parametrize("S,R", [(64, 300), (128, 130), (256, 37)])
def test_zero_tail_skip_paths(gpu, built_lib, S, R, need_rgb):
    """Fine pass on the reference resampler's fence posts (zero-width tail), with the skip on: regular tiles of S/2 samples plus
    leftover passes (one lane per ray, which read their tap back themselves) against the fp32 kernel evaluating every sample."""
    ren, rays, t_c = _scene(gpu, S, R, seed=3 * S + R)
    with torch.no_grad():
        w_c = ops.nerf_fwd(ren.nerf_coarse.packed(gpu, "fp32"), rays, t_c, tap_layer=-1, need_rgb=False, need_feat=False)["weights"]
        g = torch.Generator(device="cpu").manual_seed(S)
        t_f, flag = ops.resample(t_c, w_c, torch.rand(R, S + 1, generator=g).to(gpu), 0.01, True, want_tail_flag=True)
        ref = ops.nerf_fwd(ren.nerf_fine.packed(gpu, "fp32"), rays, t_f, tap_layer=3, need_rgb=need_rgb)
        out = ops.nerf_fwd(ren.nerf_fine.packed(gpu, "fp16x3"), rays, t_f, tap_layer=3, need_rgb=need_rgb, zero_tail=True, tail_flag=flag)
    assert float(out["weights"][:, S // 2 + 1:].abs().max()) == 0.0
    for k in ("weights", "feat", "pts", "depth", "acc") + (("rgb",) if need_rgb else ()):
        _close(out[k], ref[k], 2e-5, f"{k} S {S} rgb {need_rgb}")
