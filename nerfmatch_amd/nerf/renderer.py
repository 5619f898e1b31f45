"""NerfRenderer: the reference's rendering API (nerfmatch/nerf/renderer.py:26-333) on MI355X kernels.

Same constructor, attributes (`ret_pfeat`, `pfeat_mask`, `feat_comb`, `unnorm_scene`, sub-modules
`nerf_coarse`, `nerf_fine`, `xyz_encoder`, `dirs_encoder`, `embedding_a`) and state-dict keys as the reference,
so `load_nerf_render_from_ckpt` and the evaluator keep working.  What differs is underneath: one coarse->fine
render is 5 kernel launches (ray generation, stratified sampling, fused pass, re-sampling, fused pass) instead of
~40 eager ops per 16k-sample chunk, and nothing but the pose touches the host.

Only the validation/inference path of the mip configuration (`embedding.type == "mip"`, view directions on) is
built; training-mode outputs (`s_fine`, noise, perturb) are out of scope.

Randomness: the reference samples stochastically even at inference (render_utils.py:276, :444, :483).  Here the
two random tensors are explicit optional arguments (`t_rand`, `jitter`); when omitted they are drawn with
torch's device generator, which is statistically the same as, but not bit-identical to, the reference's CPU
stream -- parity tests pass them in.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .embedding import PositionalEncodingMIP
from .models import NeRF

F32_EPS = float(torch.finfo(torch.float32).eps)


class NerfRenderer(nn.Module):
    def __init__(self, config, num_frames=None, training=True, stop_layer=-1):
        super().__init__()
        self.training = training
        r = config.render
        self.chunksize = r.chunksize  # kept for config compatibility; the fused kernel needs no chunking
        self.use_disp, self.perturb, self.noise_std = r.use_disp, r.perturb, r.noise_std
        self.white_bg = r.white_bg or getattr(config.data, "white_bg", False)
        self.use_viewdirs = r.use_viewdirs
        self.embed_type = getattr(config.embedding, "type", "normal")
        self.bg_color = [1.0, 1.0, 1.0] if self.white_bg else None
        self.img_wh = config.data.img_wh
        self.mip_var_scale = getattr(config.embedding, "mip_var_scale", -1)
        self.out_scr = getattr(config.data, "out_scr", False)
        self.single_model = getattr(config.render, "single_model", False)
        if self.embed_type != "mip" or not self.use_viewdirs or self.out_scr or self.single_model:
            raise NotImplementedError(
                "nerfmatch_amd builds the shipped configuration only: embedding.type 'mip', use_viewdirs True, "
                "separate coarse/fine models, no scene-coordinate head (SURVEY.md section 2 row 3)")
        self.xyz_encoder = PositionalEncodingMIP(config.embedding.xyz_num_freqs)
        self.dirs_encoder = PositionalEncodingMIP(config.embedding.dirs_num_freqs)
        if config.embedding.xyz_num_freqs != 15 or config.embedding.dirs_num_freqs != 4:
            raise NotImplementedError("kernel is built for xyz_num_freqs=15, dirs_num_freqs=4")
        xyz_dim = self.xyz_encoder.get_embedding_dim(3) - 3
        dirs_dim = self.dirs_encoder.get_embedding_dim(3)
        self.appearance_embedding = getattr(config.embedding, "appearance_embed", False)
        embed_sz = 16

        def net_conf(c, stop):
            d = dict(vars(c))
            d.update(use_viewdirs=True, xyz_dim=xyz_dim, dirs_dim=dirs_dim,
                     app_dim=embed_sz if self.appearance_embedding else 0, out_3d_pnt=False, out_add_ch=0)
            if stop is not None:
                d["stop_layer"] = stop
            return d

        if getattr(config.coarse_nerf, "method", "NeRF") != "NeRF" or getattr(config.fine_nerf, "method", "NeRF") != "NeRF":
            raise NotImplementedError("only method: NeRF")
        self.num_pts_coarse = config.coarse_nerf.num_pts
        self.nerf_coarse = NeRF(net_conf(config.coarse_nerf, None))  # quirk: the coarse net never gets stop_layer
        self.num_pts_fine = config.fine_nerf.num_pts
        self.nerf_fine = NeRF(net_conf(config.fine_nerf, stop_layer))
        self.output_dim = getattr(config.fine_nerf, "output_dim", 4)
        self.embedding_a = nn.Embedding(num_frames, embed_sz) if self.appearance_embedding else None
        self.ret_pfeat = False
        self.pfeat_mask = None
        self.feat_comb = "lin"
        self.resample_padding = 0.01
        self.unnorm_scene = None
        self.last_far_fallback = None  # device int32[1] of the most recent render_novel_view
        # Arithmetic of the fused kernel's layer products:
        #   "fp16x3" (default since round 3): 16-bit matrix cores on fp16 hi/lo-split operands, three MFMAs per product, fp32
        #            accumulate: 22 mantissa bits, fp32-class results also on trained-like scenes (densities +-1e4); ~3x faster than
        #   "fp32"   v_mfma_f32_32x32x2_f32 (exact fp32 products);
        #   "bf16x3" the same split with bf16 parts (16 mantissa bits; wider exponent range): 3e-7 on smooth random-weight fields
        #            but 7e-4 on the compositing weights of the trained-like fixture -- outside the 1e-4 class there.
        self.precision = "fp16x3"
        # The fine fence posts come from the reference's randomized resampler, whose `u + u + jitter` saturates: the
        # intervals s > S/2 have zero width and therefore weight exactly 0 (NM_NERF_ZERO_TAIL in the header).  True lets
        # the bf16x3 kernel skip them -- identical outputs; False evaluates every sample like the reference does.
        self.skip_zero_tail = True
        # Arithmetic of the COARSE pass when only its compositing weights are consumed (lean=True: they feed the resampler and
        # nothing else): "same" (default since round 3) = the pass uses `precision`, the parity arithmetic; "fp16x1" = OPT-IN
        # throughput option, one fp16 MFMA per product block (a third of the matrix work, 11 significant bits: narrower than the
        # reference's fp32).  On a smooth random field the fine outputs do not notice (3.8e-7, scripts/split_precision_study.py);
        # on the trained-like fixture (sharp densities) the effect is measured by
        # tests/test_nerf_gpu.py::test_surface_fp16x1_coarse_pass_measured and stated in DESIGN.md 3.1d.  A coarse pass whose own
        # outputs are returned always uses `precision`.
        self.coarse_precision = "same"

    def set_training_mode(self, state):
        self.training = state

    # -- parameters written through `.data` behind the packed blobs --------------------------------------
    GUARD_PARAMETERS = True

    def _guard_check(self, dev):
        """One fingerprint launch per render over every parameter of the renderer (ops.ParamGuard) and, WITHOUT blocking, a look at the
        flag an earlier render left: a mismatch there means that render ran on blobs packed from older values -- warn loudly, drop the
        blobs (this render packs fresh ones).  The evaluator checks the same flag at its synchronisation point and repeats the batch
        (NeRFMatchEvaluator._localize_finish); a caller of the renderer alone can ask with check_stale()."""
        if not self.GUARD_PARAMETERS or dev.type != "cuda":
            return
        self._poll_stale(wait=False)
        g = self.__dict__.get("_guard")
        if g is None:
            g = self.__dict__["_guard"] = ops.ParamGuard(self)
        g.check()
        self._guard_publish()

    def _guard_publish(self):
        """Asynchronous copy of the guard's flag to pinned host memory behind the render just enqueued: -> token (event, host buffer), also
        kept as `_stale_last` (the evaluator takes it with the batch) and in a short list a later call of the renderer looks through."""
        g = self.__dict__.get("_guard")
        if g is None or getattr(g, "flag", None) is None:
            return None
        pool = self.__dict__.setdefault("_stale_pool", [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(4)])
        n = self.__dict__["_stale_n"] = self.__dict__.get("_stale_n", -1) + 1
        host = pool[n % 4]  # (round robin: a buffer is reused four renders later -- its copy is long complete, and the flag is sticky)
        side = ops.side_stream(g.flag.device)
        with torch.cuda.stream(side):  # (behind the fingerprint launch on the guard's own stream: the flag does not depend on the render)
            host.copy_(g.flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        tok = (ev, host)
        self.__dict__["_stale_last"] = tok
        toks = self.__dict__.setdefault("_stale_toks", [])
        toks.append(tok)
        del toks[:-4]
        return tok

    def _token_stale(self, tok, wait=True):
        """True when the render behind `tok` (or an earlier one since the last baseline) ran on blobs older than the parameters; the blobs
        are dropped then, so that the next render packs fresh ones."""
        if tok is None:
            return False
        if wait:
            tok[0].synchronize()
        elif not tok[0].query():
            return False
        if int(tok[1][1]) == 0:
            return False
        import warnings

        warnings.warn("nerfmatch_amd: NeRF parameters were modified in place through `.data` (no version bump) after their packed blobs were "
                      "made: renders since that write used the OLD values.  The blobs are rebuilt now; call invalidate() after such writes")
        self.invalidate()
        return True

    def _poll_stale(self, wait):
        toks = self.__dict__.get("_stale_toks") or []
        for tok in list(toks):
            if not wait and not tok[0].query():
                continue
            toks.remove(tok)
            if self._token_stale(tok, wait=wait):
                return True
        return False

    def check_stale(self):
        """Synchronous form: True when a render since the last check ran on blobs older than the parameters (they are rebuilt then)."""
        self._guard_publish()
        return self._poll_stale(wait=True)

    def invalidate(self):
        """Forget every copy derived from the parameters (packed blobs, operand-scale calibration, fingerprints): call after writing
        parameters through `.data`."""
        self.nerf_coarse.invalidate()
        self.nerf_fine.invalidate()
        self.__dict__.pop("_app_amax_key", None)
        self.__dict__.pop("_inerf_fused", None)  # (inerf.fused_field / fine_field: the refinement's packed copies of the fine network)
        self.__dict__.pop("_inerf_field", None)
        self.__dict__["_stale_toks"] = []
        self.__dict__["_stale_last"] = None
        if self.__dict__.get("_guard") is not None:
            self.__dict__["_guard"].reset()

    def _app_row(self, aid):
        """Row `aid` of the appearance table as the kernel's per-launch input; also tells both networks the table's maximum: the fp16x3
        scale of the appearance input covers the WHOLE table, not the row of the first batch a process happens to see."""
        w_ = self.embedding_a.weight
        key_ = (w_.data_ptr(), w_._version)
        if self.__dict__.get("_app_amax_key") != key_:
            self.__dict__["_app_amax_key"] = key_
            self.nerf_coarse.app_amax = self.nerf_fine.app_amax = float(w_.detach().abs().max())
        return w_[aid].detach().to(torch.float32).contiguous()

    def calibrate(self, device):
        """fp16x3 operand scales of both networks, chosen NOW on the seeded probe bundle (NeRF.probe_bundle) instead of lazily on the
        first render: -> {"coarse": [12 exponents], "fine": [...]} (None for other precisions).  The scales are a function of the
        parameters alone, so every rank of a sharded evaluation arrives at the same ones; dist.agree_calibration checks that."""
        if self.precision != "fp16x3":
            return None
        dev = torch.device(device)
        app_row = self._app_row(1) if self.appearance_embedding else None
        out = {}
        for name, net in (("coarse", self.nerf_coarse), ("fine", self.nerf_fine)):
            net.packed(dev, "fp32")  # (drops a calibration that belongs to earlier parameters)
            if net._act_log2.get(str(dev)) is None:
                pr, pt = NeRF.probe_bundle(dev, self.num_pts_coarse)
                net.calibrate_fp16x3(pr, pt, app_row, white_bg=self.white_bg, var_scale=self.mip_var_scale)
            out[name] = list(net._act_log2[str(dev)])
        return out

    def set_calibration(self, device, scales):
        """Adopt given fp16x3 operand scales ({"coarse": [...], "fine": [...]}; dist.agree_calibration): the blobs are re-packed on next use."""
        dev = str(torch.device(device))
        for name, net in (("coarse", self.nerf_coarse), ("fine", self.nerf_fine)):
            net._act_log2[dev] = [int(v) for v in scales[name]]
            if net._blob is not None:
                net._blob.pop((dev, "fp16x3"), None)

    # ------------------------------------------------------------------------------------------------------
    def render_rays(self, rays, ray_id=None, validation=False, t_rand=None, jitter=None, lean=False, debug=False, rgb_fine=True):
        """Coarse -> fine rendering (reference: renderer.py:182-295).

        lean=True computes only what `render_novel_view` returns (rgb_fine, pts_fine, feat_fine): the coarse pass
        then skips its colour heads and its (unused) feature sum; with rgb_fine=False the fine pass skips
        feature_linear / views / rgb as well (localisation reads only pts_fine and feat_fine: SURVEY.md 8a quirk 6).
        debug=True adds per-sample tensors."""
        if not validation:
            raise NotImplementedError("training-mode rendering (noise, s_fine/weights_fine outputs) is out of scope")
        if self.pfeat_mask is not None:
            raise NotImplementedError("pfeat_mask is a training-time option (nerf_trainer.py:45)")
        dev = rays.device
        R = rays.shape[0]
        rays = rays.to(torch.float32).contiguous()
        if rays.shape[1] < 12:
            raise ValueError("mip rendering needs the 12-column ray layout [o, d, near, far, viewdir, radius]")
        app_row = None
        if self.appearance_embedding:
            # The fused kernel takes ONE appearance row per launch.  ray_id None = the reference's default id 1 (renderer.py:298-299);
            # a tensor is inspected (on the host when it lives there, as dataset `ts` tensors do; one synchronisation when it
            # is a device tensor) and rays with different ids are rendered in one launch sequence per id.
            aid = 1
            if ray_id is not None:
                ids = torch.as_tensor(ray_id).reshape(-1)
                if ids.numel() != R:
                    raise ValueError(f"ray_id has {ids.numel()} entries for {R} rays")
                uniq = torch.unique(ids)
                if uniq.numel() > 1:
                    return self._render_rays_per_appearance(rays, ids, uniq, validation=validation, t_rand=t_rand, jitter=jitter, lean=lean,
                                                            debug=debug, rgb_fine=rgb_fine)
                aid = int(uniq[0])
            app_row = self._app_row(aid)
        # In the mip configuration the reference's resampler draws as many fence posts as it is given (resample_gaus_along_rays passes
        # t_vals.shape[-1], render_utils.py:594-597, and sample_smth_along_rays never hands it num_pts, :299-309): the fine pass has the
        # COARSE pass's sample count whatever fine_nerf.num_pts says.  Reproduced: fine_nerf.num_pts is read and not used.
        Sc = Sf = self.num_pts_coarse
        jitter_scale = 1.0
        if t_rand is None and jitter is None:  # both samplers' draws in one generator launch; the jitter's factor is applied by the resampler
            both = torch.rand(2, R, Sc + 1, device=dev)
            t_rand, jitter, jitter_scale = both[0], both[1], 1.0 / (Sf + 1) - F32_EPS
        if t_rand is None:
            t_rand = torch.rand(R, Sc + 1, device=dev)
        if jitter is None:
            jitter = torch.rand(R, Sf + 1, device=dev) * (1.0 / (Sf + 1) - F32_EPS)
        want_feat = bool(self.ret_pfeat)
        fmax = self.feat_comb == "max"
        preds = {}
        t_c = ops.sample_coarse(rays, t_rand.to(dev, torch.float32).contiguous(), Sc)
        weights_only = lean and not debug and self.precision in ("bf16x3", "fp16x3") and self.coarse_precision == "fp16x1"
        oc = self.nerf_coarse.fused("fp16x1" if weights_only else self.precision, rays, t_c, app_row, tap_layer=-1, white_bg=self.white_bg,
                          var_scale=self.mip_var_scale, need_rgb=not lean, need_feat=want_feat and not lean,
                          feat_max=fmax, want_raw=debug, want_sample_feat=debug)
        # the re-sampler reports on the device whether its output has the zero-width tail (it has for every jitter >= 0); the
        # fused kernel reads that flag and evaluates every sample if not -- no promise, no host synchronisation
        skip = bool(self.skip_zero_tail)
        t_f = ops.resample(t_c, oc["weights"], jitter.to(dev, torch.float32).contiguous(), self.resample_padding, True, want_tail_flag=skip,
                           jitter_scale=jitter_scale)
        t_f, tail_flag = t_f if skip else (t_f, None)
        of = self.nerf_fine.fused(self.precision, rays, t_f, app_row, tap_layer=self.nerf_fine.stop_layer,
                          white_bg=self.white_bg, var_scale=self.mip_var_scale, need_rgb=bool(rgb_fine) or not lean, need_feat=want_feat,
                          feat_max=fmax, want_raw=debug, want_sample_feat=debug, zero_tail=skip, tail_flag=tail_flag)
        for key, o, t in (("coarse", oc, t_c), ("fine", of, t_f)):
            if key == "coarse" and weights_only:
                continue  # (nothing but the weights of that pass is meant to be read)
            if o["feat"] is not None:
                preds[f"feat_{key}"] = o["feat"]
            preds[f"pts_{key}"] = o["pts"]
            if o["rgb"] is not None:
                preds[f"rgb_{key}"] = o["rgb"]
            preds[f"depth_{key}"] = o["depth"]
            if debug:
                preds[f"weights_{key}"], preds[f"t_{key}"], preds[f"acc_{key}"] = o["weights"], t, o["acc"]
                preds[f"raw_{key}"], preds[f"sfeat_{key}"] = o["raw"], o["sample_feat"]
        self._guard_check(dev)  # (behind the render's launches: its ~50 us of host work then sit under the kernels, not in front of the first one)
        return preds

    def _render_rays_per_appearance(self, rays, ids, uniq, t_rand=None, jitter=None, **kw):
        """Rays with different appearance ids (reference: `embedding_a(ray_id)` per ray, renderer.py:225): one render per id,
        results scattered back into ray order."""
        dev = rays.device
        out = {}
        for u in uniq.tolist():
            sel = torch.nonzero(ids == u).reshape(-1)
            sel_d = sel.to(dev)
            sub = self.render_rays(rays[sel_d].contiguous(), ray_id=torch.full((1,), u, dtype=torch.long).expand(sel.numel()),
                                   t_rand=None if t_rand is None else t_rand[sel.to(t_rand.device)],
                                   jitter=None if jitter is None else jitter[sel.to(jitter.device)], **kw)
            for k, v in sub.items():
                if k not in out:
                    out[k] = v.new_empty((rays.shape[0],) + tuple(v.shape[1:]))
                out[k][sel_d] = v
        return out

    def forward(self, rays, step=0, ray_id=None, validation=False, **kw):
        return self.render_rays(rays, ray_id, validation=validation, **kw)  # ray_id None = the default appearance id 1

    def predict(self, rays, w, h, out_raw=False, ray_id=None, **kw):
        self.set_training_mode(False)
        preds = self.forward(rays, validation=True, ray_id=ray_id, **kw)
        if out_raw:
            return preds
        for k in ("rgb_coarse", "depth_coarse", "rgb_fine", "depth_fine"):
            if k in preds and h * w == preds[k].shape[0]:
                preds[k] = preds[k].reshape(h, w, -1)
        return preds

    def render_novel_views(self, img_hw, K, c2ws, unnorm_scene, device, downsample=8, t_rand=None, jitter=None, lean=True,
                           want_im_pred=True):
        """Batched form of render_novel_view: Q world poses (Q,4,4) -> {im_pred (Q,H/ds,W/ds,3), pt3d (Q,R,3),
        pt_feat (Q,R,256)} with ONE launch per kernel over the Q*R rays (rays carry their own origin, so a batch of
        queries is just a longer ray bundle).  More workgroups per launch shrink the last partially filled round of
        the fused kernel (2400 workgroups on 256 CUs = 9.4 rounds for one 640x480 query)."""
        self.ret_pfeat = True
        H, W = int(img_hw[0]), int(img_hw[1])
        if isinstance(unnorm_scene, np.ndarray):
            unnorm_scene = torch.from_numpy(unnorm_scene)
        unnorm = unnorm_scene.detach().to("cpu", torch.float32)
        inv = torch.linalg.inv(unnorm)
        c2ws = torch.as_tensor(c2ws).detach().to("cpu", torch.float32).reshape(-1, 4, 4)
        Q = c2ws.shape[0]
        R = ops.lib().nm_raygen_count(H, W, int(downsample))
        rays, flags = ops.raygen_batch(torch.as_tensor(K), inv[None] @ c2ws, H, W, device, ds=downsample)
        self.last_far_fallback = flags
        preds = self.predict(rays, 1, 1, out_raw=True, t_rand=t_rand, jitter=jitter, lean=lean, rgb_fine=want_im_pred)
        pt3d = ops.unnormalize_points(preds["pts_fine"], unnorm)
        h, w = H // downsample, W // downsample
        im = None
        if "rgb_fine" in preds:
            im = preds["rgb_fine"].reshape(Q, h, w, 3) if h * w == R else preds["rgb_fine"].reshape(Q, R, 3)
        return dict(im_pred=im, pt3d=pt3d.reshape(Q, R, 3), pt_feat=preds["feat_fine"].reshape(Q, R, 256))

    def render_novel_view(self, img_hw, K, c2w, unnorm_scene, device, downsample=8, t_rand=None, jitter=None, lean=True,
                          want_im_pred=True):
        """World pose -> {im_pred (H/ds, W/ds, 3), pt3d (R,3) world, pt_feat (R,256)} (renderer.py:315-333).
        The 4x4 algebra (scene normalisation) is done on the host in fp32 like the reference's CPU path.
        want_im_pred=False (with lean): im_pred is None and the fine pass skips its colour heads -- what the localisation
        loop needs (it reads pt3d and pt_feat only, nerfmatch_evaluator.py:566-573)."""
        self.ret_pfeat = True
        H, W = int(img_hw[0]), int(img_hw[1])
        if isinstance(unnorm_scene, np.ndarray):
            unnorm_scene = torch.from_numpy(unnorm_scene)
        unnorm = unnorm_scene.detach().to("cpu", torch.float32)
        pose = torch.linalg.inv(unnorm) @ torch.as_tensor(c2w).detach().to("cpu", torch.float32)
        rays, flag = ops.raygen(torch.as_tensor(K), pose, H, W, device, ds=downsample)
        self.last_far_fallback = flag
        preds = self.predict(rays, W // downsample, H // downsample, t_rand=t_rand, jitter=jitter, lean=lean, rgb_fine=want_im_pred)
        pt3d = ops.unnormalize_points(preds["pts_fine"], unnorm)
        return dict(im_pred=preds.get("rgb_fine"), pt3d=pt3d, pt_feat=preds["feat_fine"])
