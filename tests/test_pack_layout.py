"""CPU: the packed weight blob + the register-resident MFMA dataflow reproduce the oracle MLP.

Runs nm_nerf_pack (host code of the C-ABI library) and replays the kernel's lane/register dataflow
with tests/mfma_emulator.py.  No GPU involved: this pins the layout logic of nerf_fwd.hip."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from nerfmatch_amd import synth, _lib
from oracle import nerf_oracle as no
import mfma_emulator as em


@pytest.mark.parametrize("case,net", [("r32_s32", "nerf_fine"), ("r128_s64_app", "nerf_coarse"), ("r128_s64_app", "nerf_fine")])
def test_packed_chain_matches_oracle(built_lib, case, net):
    fx = load_golden(f"nerf_{case}")
    sd = synth.nerf_state_dict(seed=int(fx["weights_seed"]), app_vocab=5 if fx["app"] else 0, density_bias=3.0)
    blob = _lib.pack_nerf_weights(sd, net).numpy()
    assert blob.shape[0] == em.OFF_WV + em.VK * 256
    S = fx["S"]
    ipe = fx["ipe_coarse"][:32]
    view = fx["rays"][:1, 8:11].expand(32, 3)
    dpe = no.dir_pe(view, 4)
    app = fx["app_row"] if fx["app"] else None
    tap = fx["stop_layer"] if net == "nerf_fine" else -1
    raw, feat = no.nerf_mlp(sd, net, ipe, dpe, None if app is None else app.view(1, -1).expand(32, -1), stop_layer=tap)
    sig, f, rgb = em.run_wave(blob, ipe.numpy(), dpe.numpy(), None if app is None else app.numpy(), tap if tap >= 0 else 7)
    np.testing.assert_allclose(sig, raw[:, 3].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(f, feat.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(rgb, raw[:, :3].numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("act", [None, [12, 8, 7, 7, 7, 6, 6, 6, 5, 7, 12, 3], [10, -3, 0, 2, 9, -6, 1, 4, 0, 11, 14, -2]])
@pytest.mark.parametrize("case,net,style", [("r128_s64_app", "nerf_fine", None), ("surf_w1_p21", "nerf_coarse", "surface")])
def test_fp16x3_scaled_pack_bookkeeping(built_lib, case, net, style, act):
    """Round 4: nm_nerf_pack_fp16x3_scaled's power-of-two scales cancel exactly.  The fp16x3 blob (weights x 2^a per input group,
    biases / head vectors / re-packing multipliers / tap descale in the small block) is read back and the kernel's scaled layer chain is
    replayed in float64: density, tapped features and colours must equal the oracle MLP for ANY activation exponents -- including
    ones no calibration would choose -- up to the 22-bit representation of the weights."""
    fx = load_golden(f"nerf_{case}")
    app = bool(fx["app"])
    sd = synth.nerf_state_dict(seed=int(fx["weights_seed"]), app_vocab=5 if app else 0,
                               density_bias=float(fx["density_shift"]) if style else 3.0, style=style)
    blob = _lib.pack_nerf_weights(sd, net, "fp16x3", act_log2=act).numpy()
    small, mats = em.unpack_fp16x3(blob, 16 if app else 0)
    # weights sit in the middle of the fp16 range: the largest entry of every hidden group in [2^13, 2^14)
    for name in ("pts1", "pts2", "pts3", "pts4", "pts6", "pts7"):
        assert 2.0 ** 13 <= np.abs(mats[name]).max() < 2.0 ** 14, name
    n = 48
    rays, t = fx["rays"][:4], fx["t_coarse"][:4, :13]
    mean, var = no.frustum_gaussians(t, rays[:, :3], rays[:, 3:6], rays[:, 11:12])
    ipe = no.ipe(mean.reshape(-1, 3), var.reshape(-1, 3), 15)[:n]
    dpe = no.dir_pe(rays[:1, 8:11].expand(n, 3), 4)
    app_row = fx["app_row"] if app else None
    tap = fx["stop_layer"] if net == "nerf_fine" else 7
    sd64 = {k: v.double() for k, v in sd.items()}
    raw, feat = no.nerf_mlp(sd64, net, ipe.double(), dpe.double(), None if app_row is None else app_row.double().view(1, -1).expand(n, -1),
                            stop_layer=tap if net == "nerf_fine" else -1)
    sig, f, rgb = em.run_chain_fp16x3(small, mats, ipe.numpy(), dpe.numpy(), None if app_row is None else app_row.numpy(), tap)
    scale = max(1.0, float(raw[:, 3].abs().max()))
    np.testing.assert_allclose(sig, raw[:, 3].numpy(), rtol=0, atol=3e-6 * scale)
    np.testing.assert_allclose(f, feat.numpy(), rtol=0, atol=3e-6 * max(1.0, float(feat.abs().max())))
    np.testing.assert_allclose(rgb, raw[:, :3].numpy(), rtol=0, atol=3e-6)


def test_fp16x3_pack_rejects_bad_exponents(built_lib):
    sd = synth.nerf_state_dict(seed=0, density_bias=3.0)
    with pytest.raises(_lib.NerfmatchAmdError):
        _lib.pack_nerf_weights(sd, "nerf_fine", "fp16x3", act_log2=[16] + [0] * 11)  # IPE (|x| <= 1) beyond the fp16 range
    with pytest.raises(_lib.NerfmatchAmdError):
        _lib.pack_nerf_weights(sd, "nerf_fine", "fp16x3", act_log2=[12] + [0] * 8 + [40, 12, 0])
