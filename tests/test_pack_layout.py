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
