"""CPU: host-side logic added in round 5 (no kernel calls): the calibration probe bundle, the speculative fine stage's capacity rule,
the padding rule of ops.nerf_fwd, the timing proxy's choice of entry points."""
import torch

from nerfmatch_amd import _lib, latency, synth
from nerfmatch_amd.matcher import NeRFMatcherMS
from nerfmatch_amd.nerf.models import NeRF


def test_probe_bundle_is_deterministic_and_inside_the_unit_sphere():
    NeRF._PROBES.clear()
    rays, t = NeRF.probe_bundle("cpu", 64)
    NeRF._PROBES.clear()
    rays2, t2 = NeRF.probe_bundle("cpu", 64)
    assert torch.equal(rays, rays2) and torch.equal(t, t2)  # seeded numpy PCG64: the same in every process and on every rank
    assert rays.shape == (NeRF.PROBE_RAYS, 12) and t.shape == (NeRF.PROBE_RAYS, 65)
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7]
    assert float(o.norm(dim=1).max()) <= 0.6 + 1e-6 and torch.allclose(d.norm(dim=1), torch.ones(len(d)), atol=1e-6)
    assert torch.equal(rays[:, 3:6], rays[:, 8:11]) and bool((far > near).all())
    end = o + d * far[:, None]
    assert torch.allclose(end.norm(dim=1), torch.ones(len(end)), atol=1e-5)  # the far plane is the unit sphere (scene_utils.py:101-120)
    assert bool((t[:, 1:] > t[:, :-1]).all()) and torch.allclose(t[:, 0], near) and torch.allclose(t[:, -1], far, atol=1e-6)
    assert NeRF.probe_bundle("cpu", 128)[1].shape == (NeRF.PROBE_RAYS, 129)


def test_speculative_capacity_follows_the_counts_seen():
    m = NeRFMatcherMS(synth.matcher_config("c2f"))
    assert m._spec_cap(4800) == 256                      # nothing seen yet
    m._spec_observe(150)
    assert m._spec_cap(4800) == 512                      # twice the largest count, as a power of two
    m._spec_observe(1300)
    assert m._spec_cap(4800) == 4096 and m._spec_cap(3600) == 3600  # capped by the token count
    m._spec_observe(40)
    assert m._spec_cap(4800) == 4096                     # never shrinks
    assert m._spec_cap(100) == 100
    m._spec_observe(3400)                                # round 6: a trained matcher's ~3.4 k matches of 4800 tokens -- no 4096 ceiling
    assert m._spec_cap(4800) == 4800 and m._spec_cap(16384) == 8192


def test_nerf_fwd_row_length_rule():
    """The native row length a requested samples-per-ray runs on (ops.nerf_fwd pads with zero-width intervals)."""
    native = lambda S: S if (S in (32, 64) or S % 128 == 0) else (32 if S < 32 else 64 if S < 64 else (S + 127) // 128 * 128)
    assert [native(s) for s in (20, 32, 48, 64, 96, 128, 160, 192, 256, 300)] == [32, 32, 64, 64, 128, 128, 256, 256, 256, 384]
    import inspect
    from nerfmatch_amd import ops
    assert "S_req + 127) // 128 * 128" in inspect.getsource(ops.nerf_fwd)  # (the rule tested above is the one in the wrapper)


def test_timing_proxy_times_launching_entry_points_only():
    class Fake:
        def __getattr__(self, name):
            return lambda *a: 0

    proxy = latency.TimedLib(Fake())
    raw = proxy.nm_nerf_pack_fp16x3_scaled          # host-side packing: never wrapped
    assert raw.__name__ == "<lambda>" and raw() == 0
    assert proxy.nm_abi_version.__name__ == "<lambda>"  # no stream argument
    assert proxy.nm_match_workspace_bytes.__name__ == "<lambda>"  # size query
    assert proxy.nm_layernorm.__name__ == "timed" and proxy.nm_nerf_fwd_fp16x3_ex.__name__ == "timed"
    assert set(_lib.SIGNATURES) >= {"nm_mip_encode", "nm_fourier_embed"}


def test_steady_gc_freezes_and_restores_reentrantly():
    """_lib.steady_gc: resident objects leave the cyclic collector's lists for the duration of a loop (no full collections walking them),
    nesting keeps them out until the outermost exit"""
    import gc

    from nerfmatch_amd import _lib

    assert gc.get_freeze_count() == 0
    with _lib.steady_gc():
        n = gc.get_freeze_count()
        assert n > 1000
        with _lib.steady_gc():
            pass
        assert gc.get_freeze_count() >= n  # the inner exit did not thaw the outer loop's objects
        cyc = []
        cyc.append(cyc)  # garbage made inside the loop is still collected
        del cyc
        assert gc.collect() >= 1
    assert gc.get_freeze_count() == 0
    try:
        with _lib.steady_gc():
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert gc.get_freeze_count() == 0
    # a heap the application froze itself is left alone, before and after
    gc.freeze()
    try:
        n0 = gc.get_freeze_count()
        with _lib.steady_gc():
            assert gc.get_freeze_count() == n0
        assert gc.get_freeze_count() == n0
    finally:
        gc.unfreeze()
    assert _lib._GC_DEPTH == [0]


def test_cu_mask_words_for_partition_streams():
    """Round 6: masks of the CU-partitioned streams (bit i = unit i // 8 of XCD i % 8).  A contiguous range of a multiple of 32 bits has the
    same number of units in every XCD; whole XCDs take every eighth bit; complementary partitions tile the chip."""
    ncu = 256
    bits = lambda words: {32 * w + b for w, v in enumerate(words) for b in range(32) if v >> b & 1}
    a = bits(_lib.cu_mask_words(ncu, 0, 160))
    b = bits(_lib.cu_mask_words(ncu, 160, 96))
    assert a == set(range(160)) and b == set(range(160, 256)) and not (a & b) and len(a | b) == ncu
    for part in (a, b):
        per_xcd = [sum(1 for i in part if i % 8 == x) for x in range(8)]
        assert len(set(per_xcd)) == 1  # balanced over the XCDs
    r = bits(_lib.cu_mask_words(ncu, xcds=(0, 5)))
    m = bits(_lib.cu_mask_words(ncu, xcds=(5, 3)))
    assert len(r) == 160 and len(m) == 96 and not (r & m) and len(r | m) == ncu
    assert {i % 8 for i in r} == {0, 1, 2, 3, 4} and {i % 8 for i in m} == {5, 6, 7}
    import pytest
    with pytest.raises(_lib.NerfmatchAmdError):
        _lib.cu_mask_words(ncu, 200, 96)
    with pytest.raises(_lib.NerfmatchAmdError):
        _lib.cu_mask_words(ncu, xcds=(6, 3))


def test_loop_partitions_are_off_where_they_do_not_apply():
    """The two-partition loop is used for small batches of a plain localisation loop only; on a CPU evaluator, with iterated localisation, with
    refinement or without a renderer eval_data_loader stays on one stream (NeRFMatchEvaluator._pipeline_streams)."""
    from argparse import Namespace

    from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator

    ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("coarse"), exp=Namespace(seed=0), data=Namespace()))
    assert ev.overlap_render and ev.render_part == ("xcd", 0, 5) and ev.match_part == ("xcd", 5, 3) and ev.overlap_max_queries == 4 and ev.split_step is None
    base = dict(iters=1, inerf_conf=None, retrieval_only=False, cached_pt=False, query2query=True)
    assert ev.device.type == "cpu" and ev._pipeline_streams(object(), base) is None          # no GPU
    ev.device = torch.device("cuda:0")                                                       # (pretend: the gating comes before any stream is made)
    assert ev._pipeline_streams(None, base) is None                                          # no renderer
    assert ev._pipeline_streams(object(), dict(base, iters=2)) is None                       # re-renders depend on the matcher's result
    assert ev._pipeline_streams(object(), dict(base, inerf_conf=Namespace())) is None
    assert ev._pipeline_streams(object(), dict(base, cached_pt=True, query2query=False)) is None  # cached points: nothing is rendered
    ev.overlap_render = False
    assert ev._pipeline_streams(object(), base) is None


def test_query_view_of_a_batch():
    """One-query view of a batch (iNeRF over a batch of queries): per-query inputs sliced as views, earlier match outputs not carried over."""
    from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator

    Q = 3
    batch = dict(image=torch.arange(Q * 3 * 2 * 2, dtype=torch.float32).reshape(Q, 3, 2, 2), K=torch.eye(3)[None].repeat(Q, 1, 1), c2w=torch.eye(4)[None].repeat(Q, 1, 1),
                 im_mask=torch.ones(Q, 5, dtype=torch.bool), pt2d=torch.zeros(Q, 5, 2), mpt2d_f=torch.zeros(Q, 2), match_ids=(torch.zeros(Q),) * 3,
                 _host=dict(K=torch.eye(3)[None].repeat(Q, 1, 1) * torch.arange(1, Q + 1)[:, None, None]))
    sub = NeRFMatchEvaluator._query_view(batch, 1, Q)
    assert sub["image"].shape == (1, 3, 2, 2) and torch.equal(sub["image"][0], batch["image"][1]) and sub["image"].data_ptr() == batch["image"][1].data_ptr()
    assert sub["K"].shape == (1, 3, 3) and sub["_host"]["K"].shape == (1, 3, 3) and float(sub["_host"]["K"][0, 0, 0]) == 2.0
    assert "mpt2d_f" not in sub and "match_ids" not in sub


def test_scene_cache_writer_thread_contents_and_errors(tmp_path, monkeypatch):
    """NerfEvaluator.cache_scene_pts writes its files from a worker thread behind the next group's render (round 6): every frame's file holds ITS
    rows of the group's outputs (the staging buffers are reused two groups later), in the reference's key order, with one writer or three; an
    exception in the writer surfaces in the caller."""
    from argparse import Namespace

    import numpy as np
    import torch

    from nerfmatch_amd import synth
    from nerfmatch_amd.nerf_evaluator import NerfEvaluator

    cfg = synth.nerf_config("7scenes", num_pts=32, img_wh=(64, 32))
    cfg.exp, cfg.split, cfg.downsample = Namespace(seed=0), "train", 8
    R = 32
    frames = [dict(img_wh=torch.tensor([[8, 4]]), rays=torch.full((1, R, 12), float(f)), rgbs=torch.zeros(1, R, 3), img_idx=[f"frame{f:03d}"])
              for f in range(9)]  # (no unnorm_scene: un-normalising is a device kernel, and this test runs without one)
    nev = NerfEvaluator(cfg, stop_layer=3, data_loader=frames)

    def predict(rays, w, h, out_raw=False, ray_id=None, **kw):  # frame f: points = f, features = f + column / 1000, colour = 0.25
        col = torch.arange(256, dtype=torch.float32) / 1000
        return dict(pts_fine=rays[:, :3].clone(), feat_fine=rays[:, :1] + col[None], rgb_fine=rays[:, :3] * 0 + 0.25)

    nev.model.predict = predict
    for writers in (1, 3):
        nev.cache_writers = writers
        files = nev.cache_scene_pts(cache_dir=tmp_path / f"w{writers}", frames_per_launch=2)
        assert [f.name for f in files] == [f"frame{f:03d}.npy" for f in range(9)]
        for f, path in enumerate(files):
            d = np.load(path, allow_pickle=True).item()
            assert list(d) == ["pt3d", "unnorm_scene", "pt_feat", "pt_color"]
            assert d["pt3d"].shape == (R, 3) and d["pt_feat"].shape == (R, 256) and d["pt_color"].shape == (R, 3)
            assert np.all(d["pt3d"] == f) and np.allclose(d["pt_feat"][:, 7], f + 0.007) and np.all(d["pt_color"] == 0.25)
            assert path.stat().st_size < 1.5 * (R * (3 + 256 + 3) * 4 + 2000)  # (a frame's rows, not a pickled view that drags the staging buffer along)
    real_save = np.save

    def failing_save(path, obj, *a, **k):
        if str(path).endswith("frame004.npy"):
            raise OSError("disk full (test)")
        return real_save(path, obj, *a, **k)

    monkeypatch.setattr(np, "save", failing_save)
    nev.cache_writers = 1
    try:
        nev.cache_scene_pts(cache_dir=tmp_path / "err", frames_per_launch=2)
        raised = False
    except OSError as e:
        raised = "disk full" in str(e)
    assert raised
