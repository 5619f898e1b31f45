"""Round 4: the fp16x3 kernel's safety net (VERDICT r3 item 1b / ADVICE r3 renderer.py:84).  An operand that reaches +-65504 is
clamped by the kernel (v_med3) -- and REPORTED: status[0] of nm_nerf_fwd_fp16x3_ex goes up, the guarded fp32 launch behind it
(nm_nerf_fwd_guarded, decision on the device) rewrites every output, and the managed path (NeRF.fused) re-calibrates the operand
scales.  Never a silently clamped result."""
import warnings

import pytest
import torch

from conftest import load_golden
from nerfmatch_amd import _lib, ops, synth
from test_nerf_gpu import make_renderer, maxdiff

pytestmark = pytest.mark.gpu
OUT_KEYS = ("weights", "feat", "pts", "rgb", "depth", "acc")


def scaled_copy(ren, layer, factor, gpu):
    """a second renderer whose fine network has pts_linears[layer] multiplied by `factor`"""
    import copy
    ren2 = copy.deepcopy(ren)
    with torch.no_grad():
        ren2.nerf_fine.pts_linears[layer].weight.mul_(factor)
        ren2.nerf_fine.pts_linears[layer].bias.mul_(factor)
    ren2.nerf_fine.invalidate()
    return ren2.to(gpu)


def test_saturation_raises_flag_and_fp32_pass_rewrites(gpu, built_lib):
    fx = load_golden("nerf_surf_w1_p21")
    ren, sd = make_renderer(fx, gpu)
    rays, t = fx["rays"].to(gpu), fx["t_fine"].to(gpu)
    net = ren.nerf_fine
    net.fused("fp16x3", rays, t, None, tap_layer=3)
    g0 = net.packed(gpu, "fp16x3").nm_guard
    act = list(g0.act_log2)
    sat, rng = g0.read()
    assert not sat and max(rng[:9]) < 4096.0  # calibrated: maxima in [2^10, 2^11) (+ batch-to-batch slack), >= 2^4 below the limit
    print("calibrated act_log2", act, "ranges (scaled)", [round(r) for r in rng])
    # the same scales on a network whose layer 3 is 300 x larger: layer 4's input outgrows the fp16 range at 2^act
    ren2 = scaled_copy(ren, 3, 300.0, gpu)
    sd2 = {f"m.{k}": v for k, v in ren2.nerf_fine.state_dict().items()}
    blob = _lib.pack_nerf_weights(sd2, "m", "fp16x3", act_log2=act).to(gpu)
    blob32 = ren2.nerf_fine.packed(gpu, "fp32")
    guard = ops.Fp16Guard(gpu, blob32, act)
    o = ops.nerf_fwd(blob, rays, t, None, tap_layer=3, guard=guard)
    sat, rng = guard.read()
    assert sat, "an activation 300 x beyond the calibrated range must raise the saturation flag"
    assert rng[3] >= 65504.0
    o32 = ops.nerf_fwd(blob32, rays, t, None, tap_layer=3)
    for k in OUT_KEYS:
        assert torch.equal(o[k], o32[k]), k  # the guarded fp32 launch rewrote every output
    # without the flag the fp32 pass must NOT run: outputs of the clean network stay the fp16x3 kernel's own
    o_ok = ops.nerf_fwd(net.packed(gpu, "fp16x3"), rays, t, None, tap_layer=3)
    o_ok32 = ops.nerf_fwd(net.packed(gpu, "fp32"), rays, t, None, tap_layer=3)
    assert not g0.read()[0]
    assert not torch.equal(o_ok["feat"], o_ok32["feat"])
    # the managed path calibrates the 300 x network for itself: no saturation, fp32-class results
    o2 = ren2.nerf_fine.fused("fp16x3", rays, t, None, tap_layer=3)
    g2 = ren2.nerf_fine.packed(gpu, "fp16x3").nm_guard
    assert not g2.read()[0]
    assert g2.act_log2[4] < act[4] - 6 and g2.act_log2[5] < act[5] - 6
    scale = float(o32["feat"].abs().max())
    assert maxdiff(o2["feat"], o32["feat"].cpu()) < 1e-4 * scale
    assert maxdiff(o2["weights"], o32["weights"].cpu()) < 2e-4


def test_outgrown_calibration_warns_and_recalibrates(gpu, built_lib):
    fx = load_golden("nerf_surf_w2_p23")
    ren, sd = make_renderer(fx, gpu)
    rays, t = fx["rays"].to(gpu), fx["t_fine"].to(gpu)
    net = ren.nerf_fine
    ref = ops.nerf_fwd(net.packed(gpu, "fp32"), rays, t, None, tap_layer=3)
    net.fused("fp16x3", rays, t, None, tap_layer=3)
    good = list(net._act_log2[str(gpu)])
    # pretend the calibration batch had been 2^7 quieter than the data that follows
    net._act_log2[str(gpu)] = [good[0]] + [c + 7 for c in good[1:10]] + good[10:]
    net._blob.pop((str(gpu), "fp16x3"), None)
    scale = float(ref["feat"].abs().max())
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for i in range(24):
            o = net.fused("fp16x3", rays, t, None, tap_layer=3)
            torch.cuda.synchronize()
            assert maxdiff(o["feat"], ref["feat"].cpu()) < 1e-4 * scale, i  # saturated launches were redone in fp32 on the device
    assert any("65504" in str(w.message) for w in caught), "the saturation must be reported"
    assert net._act_log2[str(gpu)] == good  # re-calibrated on the same data
    assert not net.packed(gpu, "fp16x3").nm_guard.read()[0]


def test_unguarded_fp16x3_blob_is_refused(gpu, built_lib):
    fx = load_golden("nerf_r32_s32")
    ren, sd = make_renderer(fx, gpu)
    raw = _lib.pack_nerf_weights({f"m.{k}": v for k, v in ren.nerf_fine.state_dict().items()}, "m", "fp16x3").to(gpu)
    with pytest.raises(_lib.NerfmatchAmdError):
        ops.nerf_fwd(raw, fx["rays"].to(gpu), fx["t_coarse"].to(gpu))


def test_appearance_row_scale_and_flag(gpu, built_lib):
    """Cambridge variant: the appearance row is a run-time input -- its scale comes from the calibration batch, and a row that outgrows
    it is caught by range slot 9 like an activation."""
    fx = load_golden("nerf_surface_r256_s64_app")
    ren, sd = make_renderer(fx, gpu)
    rays, t, app = fx["rays"].to(gpu), fx["t_coarse"].to(gpu), fx["app_row"].to(gpu)
    net = ren.nerf_coarse
    o = net.fused("fp16x3", rays, t, app, tap_layer=-1, white_bg=True)
    g = net.packed(gpu, "fp16x3").nm_guard
    assert not g.read()[0] and g.act_log2[11] > 0
    o32 = ops.nerf_fwd(net.packed(gpu, "fp32"), rays, t, app, tap_layer=-1, white_bg=True)
    assert maxdiff(o["rgb"], o32["rgb"].cpu()) < 1e-4
    big = app * 4096.0
    ob = ops.nerf_fwd(net.packed(gpu, "fp16x3"), rays, t, big, tap_layer=-1, white_bg=True)
    assert g.read()[0]
    ob32 = ops.nerf_fwd(net.packed(gpu, "fp32"), rays, t, big, tap_layer=-1, white_bg=True)
    assert torch.equal(ob["rgb"], ob32["rgb"])


def test_guarded_pass_walks_more_tiles_than_cus_and_consumes_the_flag(gpu, built_lib):
    """the persistent fall-back grid on MORE tiles than CUs (4800 rays x 64 samples = 2400 tiles: every workgroup walks ~9
    tiles and re-uses its LDS, including the small-parameter block other wavefronts copied) with a flag forced up by hand; every output
    must equal the plain fp32 kernel's.  The flag is consumed (bit 0 down again, event counted in status[11]) and a second guarded
    launch on the same status block leaves poisoned outputs alone."""
    fx = load_golden("nerf_r32_s32")
    R, S = 4800, 64
    ren, sd = make_renderer(fx, gpu, S=S)
    rays = fx["rays"].to(gpu).repeat(R // 32, 1).contiguous()
    t = ops.sample_coarse(rays, synth.uniform01((R, S + 1), 11).to(gpu), S)
    net = ren.nerf_fine
    blob32 = net.packed(gpu, "fp32")
    ref = ops.nerf_fwd(blob32, rays, t, None, tap_layer=3)
    import ctypes as C
    status = torch.zeros(16, dtype=torch.int32, device=gpu)
    status[0] = 1
    out = {k: torch.full_like(v, float("nan")) for k, v in ref.items() if v is not None}
    p = lambda k: _lib.dptr(out[k]) if k in out else C.c_void_p(0)

    def guarded():
        _lib.check(_lib.lib().nm_nerf_fwd_guarded(_lib.dptr(blob32), _lib.dptr(rays), _lib.dptr(t), None, R, S, 3, 0, -1.0, 0, p("weights"), p("feat"),
                                                  p("pts"), p("rgb"), p("depth"), p("acc"), None, None, _lib.dptr(status, torch.int32),
                                                  ops.stream()), "nm_nerf_fwd_guarded")

    guarded()
    for k in OUT_KEYS:
        assert torch.equal(out[k], ref[k]), k
    h = status.cpu()
    assert int(h[0]) & 1 == 0 and int(h[11]) == 1 and int(h[12]) == 0
    for v in out.values():
        v.fill_(float("nan"))
    guarded()  # flag down: must not touch the outputs
    torch.cuda.synchronize()
    assert all(bool(torch.isnan(v).all()) for v in out.values())
    assert int(status.cpu()[11]) == 1
