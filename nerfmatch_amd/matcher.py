"""NeRFMatcherMS (coarse-to-fine) and NeRFMatcherCoarse ("Mini") on MI355X kernels.

Same constructor arguments, attribute / sub-module names and state-dict keys as the reference model classes
(nerfmatch/nerfmatch_c2f_trainer.py:77-488, nerfmatch/nerfmatch_coarse_trainer.py:50-363); `forward` mutates the
batch dict in place exactly like the reference.  Inference runs fused kernels and builds no graph; the training step
(forward(training=True) / forward_with_metrics: GT-padded matches, focal + fine losses) goes through nerfmatch_amd.autograd,
whose backward passes are HIP kernels too.  Every tensor op of the reference between the backbone outputs and the match lists runs in the HIP
kernels of csrc/ (LayerNorm, fp32-MFMA linear layers, flash attention, dual-softmax matching, window gather,
fine expectation); torch is used for allocation and index plumbing only.
"""
import math

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .modules import init_backbone, init_backbone_8_2
from .modules.attention import GenericEncoderLayer, SelfAttentionBlock
from .nerf.embedding import FourierEmbedding


_side_stream = ops.side_stream  # (the small device-to-host copies that must not queue behind later work; ParamGuard's launches)


class StaleParameters(RuntimeError):
    """Raised at a forward pass's synchronisation point when ops.ParamGuard found parameter values that differ from the ones the packed
    copies were made from (a write through `.data`); the pass is repeated on fresh copies by its entry point -- never seen by callers."""


class PositionEncodingSine(nn.Module):
    """2-D sinusoidal table of LoFTR (third_party/loftr/position_encoding.py:24-43, temp_bug_fix=True): channels 0::4
    sin(x w_k), 1::4 cos(x w_k), 2::4 sin(y w_k), 3::4 cos(y w_k), 1-based positions; a non-persistent buffer."""

    def __init__(self, d_model, max_shape=(256, 256)):
        super().__init__()
        hh, ww = max_shape
        ypos = torch.arange(1, hh + 1, dtype=torch.float32)[None, :, None].expand(1, hh, ww)
        xpos = torch.arange(1, ww + 1, dtype=torch.float32)[None, None, :].expand(1, hh, ww)
        freq = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
        pe = torch.zeros(d_model, hh, ww)
        pe[0::4], pe[1::4] = torch.sin(xpos * freq), torch.cos(xpos * freq)
        pe[2::4], pe[3::4] = torch.sin(ypos * freq), torch.cos(ypos * freq)
        self.register_buffer("pe", pe.unsqueeze(0), persistent=False)


class FinePreprocess(nn.Module):
    """Holds `down_proj` / `merge_feat` for checkpoint compatibility: the reference computes their output and
    discards it (third_party/loftr/fine_matching.py:58-71 returns the un-merged windows), so no kernel uses them."""

    def __init__(self, win_sz=5, stride=4, d_model_f=128, d_model_c=256, cat_c_feat=True):
        super().__init__()
        self.W, self.stride, self.cat_c_feat, self.d_model_f = win_sz, stride, cat_c_feat, d_model_f
        if cat_c_feat:
            self.down_proj = nn.Linear(d_model_c, d_model_f, bias=True)
            self.merge_feat = nn.Linear(2 * d_model_f, d_model_f, bias=True)


class _MatcherBase(nn.Module):
    def _init_common(self, config):
        self.cfeat_dim = getattr(config, "cfeat_dim", 256)
        self.temp_type = getattr(config, "temp_type", "mul")
        if self.temp_type == "div":
            self.temperature = nn.Parameter(torch.tensor(0.1), requires_grad=False)
        elif self.temp_type == "mul":
            self.temperature = nn.Parameter(torch.tensor(10.0), requires_grad=True)
        else:
            raise ValueError(self.temp_type)
        self.im_pe = PositionEncodingSine(self.cfeat_dim) if getattr(config, "im_pe", True) else None
        pt_pe = getattr(config, "pt_pe", True)
        self.post_pt_pe = getattr(config, "post_pt_pe", False)
        self.pt_dim = getattr(config, "pt_dim", self.cfeat_dim)
        self.pt_ftype = getattr(config, "pt_ftype", "nerf")
        # what a 3-D point is described by (c2f_trainer.py:121-139, coarse_trainer.py:91-112): "nerf" rendered features (the shipped
        # configs), "pe3d" the 15-frequency Fourier embedding of its coordinates, "pt3d" the coordinates themselves, "rand" noise
        if self.pt_ftype not in ("nerf", "pe3d", "pt3d", "rand"):
            raise ValueError(f"pt_ftype {self.pt_ftype!r}")
        self.pt_proj = None
        if self.pt_ftype == "pe3d":
            self.pt_enc = FourierEmbedding(15)
            self.pt_dim = self.pt_enc.get_embedding_dim(3)
        elif self.pt_ftype == "pt3d":
            self.pt_dim = 3
        if self.pt_dim == 3 and self.pt_ftype != "pt3d":
            raise ValueError("pt_dim 3 means pt_ftype 'pt3d' (the reference asserts it)")
        if self.pt_dim != self.cfeat_dim:
            self.pt_proj = nn.Linear(self.pt_dim, self.cfeat_dim, bias=True)
        self.pt_pe_dim = 0
        if pt_pe:
            self.pt_pe_type = getattr(config, "pt_pe_type", "fourier")
            if self.pt_pe_type == "id":  # the "encoding" is the point's own input description, concatenated behind the self-attention
                if not self.post_pt_pe:
                    raise ValueError("pt_pe_type 'id' needs post_pt_pe (the reference asserts it)")
                self.pt_pe_dim = self.pt_dim
            else:
                self.pt_pe = FourierEmbedding(15)
                self.pt_pe_dim = self.pt_pe.get_embedding_dim(3)
            self.pt_pe_proj = nn.Linear(self.cfeat_dim + self.pt_pe_dim, self.cfeat_dim)
        self.pt_feat_normalize = False  # (NeRFMatcherCoarse reads `pt_feat_norm`)
        pt_sa_type = getattr(config, "pt_sa_type", "full")
        pt_sa = getattr(config, "pt_sa", 3)
        self.pt_sa = None
        if pt_sa_type == "full" and pt_sa > 0:
            self.pt_sa = SelfAttentionBlock(pt_sa, self.cfeat_dim, att_type="full", head_dim=self.cfeat_dim // 8)
        im_sa_type = getattr(config, "im_sa_type", None)
        im_sa = getattr(config, "im_sa", 3)
        self.im_sa = None
        if im_sa_type is not None and im_sa > 0:
            if im_sa_type == "share":
                self.im_sa = self.pt_sa  # same module object: state-dict keys im_sa.* alias pt_sa.*
            elif im_sa_type == "full":
                self.im_sa = SelfAttentionBlock(im_sa, self.cfeat_dim, att_type="full", head_dim=self.cfeat_dim // 8)
        self.cformer_type = getattr(config, "cformer_type", "crs")
        self.coarse_layers = getattr(config, "coarse_layers", 1)
        self.coarse_former = None
        if self.cformer_type.startswith("crs") and self.coarse_layers > 0:
            self.coarse_former = GenericEncoderLayer(model_dim=self.cfeat_dim, context_dim=self.cfeat_dim,
                                                     head_dim=self.cfeat_dim // 8, att_mode="cross", att_type="full")

    # -- helpers ------------------------------------------------------------------------------------------------
    def _match_scale(self):
        """Host value of the learned temperature, read back once per parameter version (a device read-back is a full
        synchronisation; four of them per step left the GPU idle between batch elements)."""
        p = self.temperature
        key = (p.data_ptr(), p._version)
        hit = self.__dict__.get("_temp_host")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_temp_host"] = (key, float(p.detach()))
        t = hit[1]
        return t if self.temp_type == "mul" else 1.0 / t

    def invalidate(self):
        """Forget the derived host / device copies (temperature read-back, padded pt_pe_proj weight, packed GEMM blobs): call
        after writing parameters through `.data` (e.g. `temperature.data.clamp_`), which does not bump `_version`."""
        self.__dict__.pop("_temp_host", None)
        self.__dict__.pop("_w_pad", None)
        for m in self.modules():
            m.__dict__.pop("_fused", None)  # (MultiHeadAttention's stacked projection weights)
        if self.__dict__.get("_guard") is not None:
            self.__dict__["_guard"].reset()
        ops.invalidate_caches()

    GUARD_PARAMETERS = True  # one ~5 us launch per inference pass that makes writes through `.data` impossible to miss (ops.ParamGuard)

    def _guard_flag(self, dev):
        """Enqueue the parameter fingerprint check of this pass; -> the device flag to read back with the match counts (None: off / CPU)."""
        if not self.GUARD_PARAMETERS or dev.type != "cuda":
            return None
        g = self.__dict__.get("_guard")
        if g is None:
            g = self.__dict__["_guard"] = ops.ParamGuard(self)
        g.check()
        return g.flag

    def _guard_early(self, t):
        """Launch this pass's fingerprint check now (t: any input tensor of the pass, for its device); coarse_match_begin collects the flag."""
        if isinstance(t, torch.Tensor) and t.is_cuda and not ag.is_training():
            self.__dict__["_guard_pending"] = self._guard_flag(t.device)

    def _retry_if_stale(self, fn):
        """fn() with ONE repetition on fresh derived copies when its read-back reports stale ones."""
        try:
            return fn()
        except StaleParameters:
            import warnings

            warnings.warn("nerfmatch_amd: matcher parameters were modified in place through `.data` (no version bump) after their packed copies "
                          "were made; the copies were rebuilt and this pass repeated -- call invalidate() after such writes to avoid the repetition")
            self.invalidate()
            return fn()

    def _padded_weight(self, lin):
        """lin.weight (N, K) zero-padded along K to a multiple of 8 (nm_linear's K granularity), cached per parameter state."""
        w = lin.weight
        if w.shape[1] % 8 == 0:
            return w
        key = (w.data_ptr(), w._version, str(w.device))
        cache = self.__dict__.setdefault("_w_pad", {})
        hit = cache.get(id(lin))
        if hit is None or hit[0] != key:
            k = w.shape[1]
            wp = torch.zeros(w.shape[0], (k + 7) // 8 * 8, device=w.device, dtype=torch.float32)
            wp[:, :k] = w.detach()
            hit = cache[id(lin)] = (key, wp)
        return hit[1]

    def _linear_any_k(self, x, lin):
        """lin(x) for any input width: x and the weight are zero-padded to the next multiple of 8 columns (exact: the padding multiplies zeros)."""
        k = x.shape[-1]
        if ag.is_training():
            kp = (k + 7) // 8 * 8
            xp = x if kp == k else torch.nn.functional.pad(x, (0, kp - k))
            # (frozen weight: the cached padded copy -- a fresh pad per call is a launch AND a miss of the packed / transposed blob caches)
            wp = lin.weight if kp == k else (torch.nn.functional.pad(lin.weight, (0, kp - k)) if lin.weight.requires_grad else self._padded_weight(lin))
            return ag.linear(xp.reshape(-1, kp), wp, lin.bias).reshape(*x.shape[:-1], -1)
        w = self._padded_weight(lin)
        if w.shape[1] != k:
            x = torch.nn.functional.pad(x, (0, w.shape[1] - k))
        return ops.linear(x.contiguous(), w, lin.bias)

    def cat_pe(self, pt_feat, pt3d, pt_feat_in=None):
        """pt_pe_proj(cat[pt_feat, encoding]) (reference :258-261): the Fourier embedding of pt3d, or -- pt_pe_type "id" -- the points' own
        input description `pt_feat_in`."""
        b, n, c = pt_feat.shape
        if self.pt_pe_type == "id":
            return self._linear_any_k(torch.cat([pt_feat, pt_feat_in.to(pt_feat.dtype)], -1), self.pt_pe_proj)
        if ag.is_training():
            cat = ag.cat_fourier(pt_feat.reshape(-1, c), pt3d.reshape(-1, 3), 15)
            w = self.pt_pe_proj.weight
            # zero columns for the zero padding of `cat` (frozen weight: the cached padded copy, see _linear_any_k)
            w_pad = torch.nn.functional.pad(w, (0, cat.shape[1] - w.shape[1])) if w.requires_grad else self._padded_weight(self.pt_pe_proj)
            return ag.linear(cat, w_pad, self.pt_pe_proj.bias).reshape(b, n, -1)
        cat = ops.cat_fourier(pt_feat.reshape(-1, c).contiguous(), pt3d.reshape(-1, 3).contiguous(), 15)
        return ops.linear(cat, self._padded_weight(self.pt_pe_proj), self.pt_pe_proj.bias).reshape(b, n, -1)

    def tokens_from_cfeat(self, cfeat, self_attention=True):
        cfeat = cfeat.to(torch.float32).contiguous()
        if ag.is_training():
            pe = self.im_pe.pe[0].contiguous() if self.im_pe is not None else None
            if self.cfeat_proj is not None:
                tok = ag.linear(ag.tokens_from_map(cfeat), self.cfeat_proj.weight, self.cfeat_proj.bias)
                if pe is not None:
                    b, c, h, w = cfeat.shape
                    tok = tok + pe[:, :h, :w].flatten(-2).T[None]
            else:
                tok = ag.tokens_from_map(cfeat, pe)
            return self.im_sa(tok) if self.im_sa is not None else tok
        if self.cfeat_proj is not None:
            tok = ops.linear(ops.nchw_to_tokens(cfeat), self.cfeat_proj.weight, self.cfeat_proj.bias)
            if self.im_pe is not None:
                b, c, h, w = cfeat.shape
                tok = tok + self.im_pe.pe[0, :, :h, :w].flatten(-2).T[None]
        else:
            tok = ops.nchw_to_tokens(cfeat, self.im_pe.pe[0].contiguous() if self.im_pe is not None else None)
        if self.im_sa is not None and self_attention:
            tok = self.im_sa(tok)
        return tok

    def _point_description(self, pt_feat, pt3d):
        """The per-point input of the encoder for the configured `pt_ftype` (reference :264-269)."""
        if self.pt_ftype == "pt3d":
            return pt3d
        if self.pt_ftype == "rand":
            return torch.randn(pt3d.shape[0], pt3d.shape[1], self.pt_dim, device=pt3d.device, dtype=torch.float32)
        if self.pt_ftype == "pe3d":
            if ag.is_training():  # (the autograd form carries the gradient to pt3d: the Fourier columns of cat_fourier, without its feature block)
                flat = pt3d.reshape(-1, 3)
                return ag.cat_fourier(flat.detach(), flat, 15)[:, 3:3 + self.pt_dim].reshape(*pt3d.shape[:-1], self.pt_dim)
            return self.pt_enc(pt3d)
        return pt_feat

    def extract_pt_feat(self, pt_feat, pt3d, im_tokens=None):
        """Point tokens (reference :263-287).  `im_tokens` (inference, `im_sa_type: share`): image tokens that have NOT been
        through the self-attention block yet and have the point tokens' shape -- both sets then go through the shared block as
        ONE batch of 2B sequences (half the launches, fuller grids; every kernel of the block works per row / per sequence, so
        the values are those of two separate calls) and (point tokens, image tokens) is returned."""
        pt3d = pt3d.to(torch.float32).contiguous()
        if self.pt_feat_normalize:
            pt_feat, pt3d = self._feature_normalization(pt_feat), self._feature_normalization(pt3d)
        pt_feat = self._point_description(None if pt_feat is None else pt_feat.to(torch.float32).contiguous(), pt3d)
        pt_feat_in = pt_feat
        if self.pt_proj is not None:
            pt_feat = self._linear_any_k(pt_feat, self.pt_proj)
        if self.pt_pe_dim > 0 and not self.post_pt_pe:
            pt_feat = self.cat_pe(pt_feat, pt3d, pt_feat_in)
        if im_tokens is not None:
            B = pt_feat.shape[0]
            both = self.pt_sa(torch.cat([im_tokens, pt_feat], 0))
            im_tokens, pt_feat = both[:B], both[B:]
        elif self.pt_sa is not None:
            pt_feat = self.pt_sa(pt_feat)
        if self.pt_pe_dim > 0 and self.post_pt_pe:
            pt_feat = self.cat_pe(pt_feat, pt3d, pt_feat_in)
        return pt_feat if im_tokens is None else (pt_feat, im_tokens)

    def _feature_normalization(self, x):
        """feature_normalization of the coarse model's `pt_feat_norm` option (coarse_trainer.py:42-47): the set is centred IN PLACE -- the
        caller's tensor changes, as in the reference -- and a copy scaled by the largest row norm is returned (nm_feature_normalize)."""
        if x.requires_grad:
            raise NotImplementedError("pt_feat_norm has no backward pass (its inputs are data in every call of the reference)")
        if not (x.is_contiguous() and x.dtype == torch.float32):
            raise ValueError("pt_feat_norm centres its input in place: hand over a contiguous fp32 tensor")
        return ops.feature_normalize(x)

    def _shared_sa_batchable(self, cfeat, pt_feat):
        """True when the image and point tokens can share one pass through the self-attention block (both are cfeat_dim wide by then)."""
        return (self.im_sa is not None and self.im_sa is self.pt_sa and not ag.is_training() and cfeat.dim() == 4 and
                cfeat.shape[0] == pt_feat.shape[0] and cfeat.shape[2] * cfeat.shape[3] == pt_feat.shape[1] and
                (self.cfeat_proj.weight.shape[0] if self.cfeat_proj is not None else cfeat.shape[1]) == self.cfeat_dim)

    def image_tokens(self, img):
        """The image side of the coarse matcher -- backbone, tokens, sine PE, self-attention block -- as a constant: with frozen parameters
        and an image that asks for no gradient nothing here is differentiated, so it runs on the fused inference kernels and builds no
        graph.  It does not depend on the rendered points either: the iNeRF refinement evaluates it ONCE per query and hands it to every
        step's match_loss (the reference recomputes it in each of its `num_optim` steps, nerfmatch_evaluator.py:429-437)."""
        if any(p.requires_grad for p in self.parameters()) or (isinstance(img, torch.Tensor) and img.requires_grad):
            return None
        with torch.no_grad(), ag.training(False):
            im = self.extract_im_feat(img)
        return (im[0] if isinstance(im, tuple) else im).detach()

    def match_loss(self, img, pt_feat, pt3d, im_mask, pt_mask, conf_gt, alpha=0.25, gamma=2.0, im_tokens=None):
        """compute_matching_loss(forward_match(...)["conf_matrix"], conf_gt) (utils/metrics.py:372-380) as a scalar that carries
        the autograd graph back to `pt_feat` / `pt3d` (and the parameters, when they require it): what the iNeRF refinement
        differentiates (nerfmatch_evaluator.py:429-441) -- either model class.  Must run inside autograd.training(); the fine stage, which
        does not enter this loss, is not evaluated."""
        if im_tokens is not None:  # (image_tokens(img), evaluated once by the caller)
            im_cfeat = im_tokens
        else:
            im_cfeat = self.extract_im_feat(img)
            if isinstance(im_cfeat, tuple):
                im_cfeat = im_cfeat[0]
        pt_cfeat = self.extract_pt_feat(pt_feat, pt3d)
        im_cfeat, pt_cfeat = self.cross(im_cfeat, pt_cfeat)
        return ag.coarse_match_loss(im_cfeat, pt_cfeat, self.temperature, self._match_scale(), im_mask, pt_mask, conf_gt, self.temp_type, True,
                                    0.0, alpha, gamma, loss_only=True)[0]

    def cross(self, im, pt):
        if self.coarse_former is None:
            return im, pt
        if self.cformer_type == "crs":  # sequential: the point side attends to the UPDATED image tokens
            im = self.coarse_former(im, pt)
            pt = self.coarse_former(pt, im)
        elif self.cformer_type == "crsv2":
            im, pt = self.coarse_former(im, pt), self.coarse_former(pt, im)
        else:
            raise NotImplementedError(self.cformer_type)
        return im, pt

    def coarse_match_begin(self, im, pt, im_mask, pt_mask, mutual, match_thres, ret_feats, keep_conf=True):
        """Enqueues the dual-softmax matching of every batch element; nothing is read back yet (see coarse_match_finish)."""
        r = ops.dual_softmax_match_batch(im, pt, self._match_scale(), im_mask, pt_mask, threshold=match_thres, mutual=mutual,
                                         want_conf=keep_conf, want_norm=ret_feats)
        # The match counts travel to pinned host memory on a side stream that waits only for the kernels enqueued so far: the
        # read-back in coarse_match_finish then does not wait for work the caller queues in between (the next batch's render),
        # and the GPU still has that work to do while the host issues the fine stage.
        dev = im.device
        # the fingerprint check of this pass: launched at the pass's START on the side stream (_guard_early: it then runs beside the encoder
        # layers), or here if the entry point did not
        flag = self.__dict__.pop("_guard_pending", None)
        if flag is None:
            flag = self._guard_flag(dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        side = _side_stream(dev)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            host = torch.empty(r["count"].numel() + 2, dtype=torch.int32, pin_memory=True)
            host[: r["count"].numel()].copy_(r["count"].reshape(-1), non_blocking=True)
            if flag is not None:
                host[r["count"].numel():].copy_(flag, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        return dict(res=r, conf=r["conf"], feats=(r["im_norm"], r["pt_norm"]) if ret_feats else None, dev=dev, count_host=host,
                    count_done=done, guarded=flag is not None)

    @staticmethod
    def coarse_match_finish(st):
        """The ONE device synchronisation of a forward pass: match counts of the batch -> (match_ids, mconf, conf, feats, counts)."""
        r = st["res"]
        st["count_done"].synchronize()
        counts = st["count_host"].tolist()
        stale = counts.pop()  # (the last two entries: ParamGuard's flag block [completion counter, stale])
        counts.pop()
        if st.get("guarded") and stale:
            raise StaleParameters()
        if len(counts) == 1:  # single pair: the valid prefix as views -- not one launch behind the read-back
            k = counts[0]
            spec = st.get("spec")
            b0 = spec["b_ids"][:k] if (spec is not None and k <= spec["cap"]) else torch.zeros(k, device=st["dev"], dtype=torch.int64)
            return (b0, r["i_ids"][0, :k], r["j_ids"][0, :k]), r["mconf"][0, :k], st["conf"], st["feats"], counts
        bs, is_, js, cs = [], [], [], []
        for b, k in enumerate(counts):
            bs.append(torch.full((k,), b, device=st["dev"], dtype=torch.int64))
            is_.append(r["i_ids"][b, :k]); js.append(r["j_ids"][b, :k]); cs.append(r["mconf"][b, :k])
        ids = (torch.cat(bs), torch.cat(is_), torch.cat(js))
        return ids, torch.cat(cs), st["conf"], st["feats"], counts

    def coarse_match(self, im, pt, im_mask, pt_mask, mutual, match_thres, ret_feats, keep_conf=True):
        """Per batch element dual-softmax matching; returns the reference's (match_ids, mconf, conf_matrix, feats)."""
        return self.coarse_match_finish(self.coarse_match_begin(im, pt, im_mask, pt_mask, mutual, match_thres, ret_feats, keep_conf))[:4]


class NeRFMatcherMS(_MatcherBase):
    def __init__(self, config):
        super().__init__()
        self.coarse_ds, self.fine_ds = 8, 2
        self.backbone = init_backbone_8_2(config.backbone, pretrained=getattr(config, "pretrained", False))
        self._init_common(config)
        self.ffeat_dim = getattr(config, "ffeat_dim", 128)
        bd = self.backbone.feat_dim
        self.cfeat_proj = nn.Linear(bd[0], self.cfeat_dim, bias=True) if bd[0] != self.cfeat_dim else None
        self.ffeat_proj = nn.Linear(bd[1], self.ffeat_dim, bias=True) if bd[1] != self.ffeat_dim else None
        self.pt_ffeat_proj = nn.Sequential(nn.Linear(self.cfeat_dim, self.ffeat_dim), nn.Linear(self.ffeat_dim, self.ffeat_dim))
        self.coarse_percent = getattr(config, "coarse_percent", 0.3)
        self.coarse_dthres = getattr(config, "coarse_dthres", 20)
        self.fine_loss = getattr(config, "fine_loss", "match")
        self.win_sz = int(getattr(config, "win_sz", 5))
        self.cat_c_feat = getattr(config, "cat_c_feat", True)
        self.fine_preprocess = FinePreprocess(win_sz=self.win_sz, stride=4, d_model_f=self.ffeat_dim, d_model_c=self.cfeat_dim,
                                              cat_c_feat=self.cat_c_feat)
        fsa_type = getattr(config, "fsa_type", "full")
        if fsa_type in ("full", "lsa"):
            self.fine_sa = SelfAttentionBlock(config.fine_sa, self.ffeat_dim, att_type=fsa_type, head_dim=self.ffeat_dim // 8)
        self.keep_conf = True  # the reference always returns conf_matrix; set False to skip its 4*M*N-byte write

    def extract_im_feat(self, img, pt_feat=None):
        """Image tokens and fine map (reference :237-256).  With `pt_feat` (the point features about to be matched): if the two
        token sets can share one pass through the self-attention block (_shared_sa_batchable) the image tokens are returned
        WITHOUT it, for extract_pt_feat(..., im_tokens=...) to finish; the third return value says which."""
        cfeat, ffeat = self.backbone(img)
        defer = pt_feat is not None and self._shared_sa_batchable(cfeat, pt_feat)
        if self.ffeat_proj is not None:
            b, f, hf, wf = ffeat.shape
            if ag.is_training():
                ff = ag.linear(ag.tokens_from_map(ffeat.contiguous()), self.ffeat_proj.weight, self.ffeat_proj.bias)
            else:
                ff = ops.linear(ops.nchw_to_tokens(ffeat.contiguous()), self.ffeat_proj.weight, self.ffeat_proj.bias)
            ffeat = ff.reshape(b, hf, wf, -1).permute(0, 3, 1, 2).contiguous()
        tok, ffeat = self.tokens_from_cfeat(cfeat, self_attention=not defer), ffeat.to(torch.float32).contiguous()
        return (tok, ffeat) if pt_feat is None else (tok, ffeat, defer)

    def forward_match(self, img, pt_feat, pt3d, im_mask=None, pt_mask=None, conf_gt=None, ret_feats=False, mutual=False,
                      match_thres=0.0):
        """reference :302-369.  With `conf_gt` (the training call, also made by the reference's iNeRF match loss) the
        predicted matches are padded with ground-truth pairs (extract_matches.py:38-56) and the pass runs through the
        autograd functions: `coarse_loss` (the focal loss of conf_matrix against conf_gt, evaluated by the kernels that hold
        the similarity matrix) and `expec_f` carry a graph whose backward is HIP kernels; `conf_matrix` itself is a value."""
        if conf_gt is not None:
            with ag.training():
                return self._train_preds(img, pt_feat, pt3d, im_mask, pt_mask, conf_gt, ret_feats=ret_feats, mutual=mutual,
                                         match_thres=match_thres)
        return self._retry_if_stale(lambda: self.forward_match_finish(self.forward_match_begin(img, pt_feat, pt3d, im_mask, pt_mask, ret_feats, mutual,
                                                                                              match_thres)))

    def forward_match_begin(self, img, pt_feat, pt3d, im_mask=None, pt_mask=None, ret_feats=False, mutual=False, match_thres=0.0):
        """Everything of forward_match up to (not including) the read-back of the match counts: encoders, cross attention and
        the dual-softmax kernels are enqueued, the returned state is completed by forward_match_finish.  A caller that has
        more GPU work to issue (the next query batch's render) does so between the two halves, which keeps the GPU busy
        across the one synchronisation point of the pipeline."""
        self._guard_early(pt3d)
        im_cfeat, im_ffeat, deferred = self.extract_im_feat(img, pt_feat)
        if deferred:
            pt_cfeat, im_cfeat = self.extract_pt_feat(pt_feat, pt3d, im_tokens=im_cfeat)
        else:
            pt_cfeat = self.extract_pt_feat(pt_feat, pt3d)
        return self._match_tokens_begin(im_cfeat, im_ffeat, pt_cfeat, im_mask, pt_mask, ret_feats, mutual, match_thres)

    def forward_match_finish(self, st):
        return self._match_tokens_finish(st)

    def _match_tokens(self, im_cfeat, im_ffeat, pt_cfeat, im_mask, pt_mask, ret_feats, mutual, match_thres, ffeat_of=None):
        """Cross attention -> dual-softmax matching -> fine stage for token batches of equal size B'.
        `ffeat_of[b']` maps a token-batch row to the row of `im_ffeat` it belongs to (multi-pair: several point sets
        share one image)."""
        return self._match_tokens_finish(self._match_tokens_begin(im_cfeat, im_ffeat, pt_cfeat, im_mask, pt_mask, ret_feats, mutual,
                                                                  match_thres, ffeat_of))

    def _match_tokens_begin(self, im_cfeat, im_ffeat, pt_cfeat, im_mask, pt_mask, ret_feats, mutual, match_thres, ffeat_of=None):
        im_cfeat, pt_cfeat = self.cross(im_cfeat, pt_cfeat)
        st = self.coarse_match_begin(im_cfeat, pt_cfeat, im_mask, pt_mask, mutual, match_thres, ret_feats, self.keep_conf)
        st.update(im_cfeat=im_cfeat, im_ffeat=im_ffeat, pt_cfeat=pt_cfeat, ret_feats=ret_feats, ffeat_of=ffeat_of)
        return st

    def _match_tokens_finish(self, st):
        im_cfeat, im_ffeat, pt_cfeat, ret_feats, ffeat_of = st["im_cfeat"], st["im_ffeat"], st["pt_cfeat"], st["ret_feats"], st["ffeat_of"]
        ids, mconf, conf, feats, counts = self.coarse_match_finish(st)
        b_ids, i_ids, j_ids = ids
        K = b_ids.shape[0]
        dev = im_cfeat.device
        spec = st.get("spec")
        if spec is not None and K <= spec["cap"]:
            # the fine stage ran on the first `cap` slots of the match list before the count was known (_speculate): take the valid prefix
            expec_f = spec["expec_f"][:K]
        elif K == 0:
            expec_f = torch.empty(0, 3, device=dev)
        else:
            # total match count as a DEVICE tensor computed on the device: torch.tensor([K], device=...) would be a pageable
            # host-to-device copy, i.e. a wait for everything the caller has queued behind this batch (the next batch's render
            # and matcher) before the fine stage could even be issued
            cnt = st["res"]["count"].sum(dtype=torch.int32).reshape(1)
            expec_f = self._fine_stage(pt_cfeat, im_ffeat, b_ids, i_ids, j_ids, cnt, ffeat_of)
            if spec is not None:
                self.__dict__["spec_reruns"] = self.__dict__.get("spec_reruns", 0) + 1  # (more matches than the speculative capacity)
        if spec is not None:
            self.__dict__["spec_batches"] = self.__dict__.get("spec_batches", 0) + 1
            self._spec_observe(K)
        pred_mask = spec["pred_mask"][:K] if (spec is not None and K <= spec["cap"]) else mconf != 0
        preds = dict(conf_matrix=conf, expec_f=expec_f, match_ids=ids, mconf=mconf, pred_mask=pred_mask, pred_num=K,
                     match_counts=counts)  # per token-batch row, host ints (read back at the synchronisation point)
        if ret_feats:
            preds.update(im_cfeat=feats[0], pt_cfeat=feats[1])
        return preds

    def _fine_stage(self, pt_cfeat, im_ffeat, b_ids, i_ids, j_ids, cnt, ffeat_of=None):
        """Fine stage for the matches (b_ids, i_ids, j_ids) -- K slots of which the first cnt[0] (device int32) are valid; the kernels
        that gather skip the rest -> expec_f (K, 3).  reference: pt_ffeat_proj + FinePreprocess + fine_sa + FineMatching,
        c2f_trainer.py:344-350, third_party/loftr/fine_matching.py:34-121."""
        B, N, C = pt_cfeat.shape
        dev = pt_cfeat.device
        # (one pair: every batch index is 0 -- no index arithmetic, two elementwise launches less on the one-query path)
        flat_j = j_ids.contiguous() if B == 1 else (b_ids * N + j_ids).contiguous()
        if (ops.FINE_STAGE_ONE_LAUNCH and ops.fine_pt_proj_supported(self.pt_ffeat_proj[0], self.pt_ffeat_proj[1])
                and ops.fine_window_layer_supported(self.fine_sa, self.win_sz, im_ffeat.shape[1])):
            # the whole fine stage -- point projection, window gather, encoder layer, expectation -- in ONE launch (nm_fine_stage)
            map_ids = b_ids if ffeat_of is None else torch.as_tensor(ffeat_of, device=dev, dtype=torch.int64)[b_ids]
            return ops.fine_window_layer(im_ffeat, map_ids, i_ids, cnt, self.fine_sa, 4,
                                         pt_proj=(pt_cfeat.reshape(B * N, C), flat_j, self.pt_ffeat_proj[0], self.pt_ffeat_proj[1]))
        if ops.fine_pt_proj_supported(self.pt_ffeat_proj[0], self.pt_ffeat_proj[1]):
            pf = ops.fine_pt_proj(pt_cfeat.reshape(B * N, C), flat_j, cnt, self.pt_ffeat_proj[0], self.pt_ffeat_proj[1])  # one launch
        else:
            pf = ops.gather_rows(pt_cfeat.reshape(B * N, C), flat_j, cnt)
            pf = ops.linear(pf, self.pt_ffeat_proj[0].weight, self.pt_ffeat_proj[0].bias)
            pf = ops.linear(pf, self.pt_ffeat_proj[1].weight, self.pt_ffeat_proj[1].bias)
        # one launch for the windows of the whole batch (match k reads the fine map of its batch row; multi-pair: of the
        # image its token-batch row belongs to)
        map_ids = b_ids if ffeat_of is None else torch.as_tensor(ffeat_of, device=dev, dtype=torch.int64)[b_ids]
        if ops.fine_window_layer_supported(self.fine_sa, self.win_sz, im_ffeat.shape[1]) and pf.shape[1] == 128:
            # window gather + the encoder layer + FineMatching's expectation: one launch
            return ops.fine_window_layer(im_ffeat, map_ids, i_ids, cnt, self.fine_sa, 4, pt_f=pf)
        else:
            win = ops.fine_windows_batch(im_ffeat, map_ids.contiguous(), i_ids.contiguous(), cnt, self.win_sz, 4)
            win = self.fine_sa(win)
        return ops.fine_expectation(pf, win, cnt, self.win_sz)

    # Single-pair batches (ONE query per step: the reference's operating point, nerfmatch_evaluator.py:631-724): the host has nothing
    # to overlap the match-count read-back with, and the ~30 small launches behind it (fine stage, assembly) would each wait for the
    # host.  So they are issued BEFORE the read-back on the first `cap` slots of the (zero-initialised) match list -- the gather kernels
    # skip slots >= count on the device -- and the read-back only slices.  cap follows the counts seen so far (twice the largest, a power
    # of two from 256 up, at most the token count); a batch with more matches than cap re-runs the fine stage the ordinary way: the result is the same.
    SPECULATE_SINGLE_PAIR = True

    def _spec_cap(self, M):
        top = self.__dict__.get("_spec_top", 64)
        cap = 256
        while cap < 2 * top and cap < M:  # (round 6: no fixed 4096 ceiling -- a trained matcher's 3-4 k matches of 4800 tokens re-ran the fine stage every batch)
            cap *= 2
        return min(cap, M)

    def _spec_observe(self, K):
        self.__dict__["_spec_top"] = max(self.__dict__.get("_spec_top", 64), int(K))

    def _speculate(self, st, pt2d=None, pt3d=None):
        """Enqueue the fine stage (and, with pt2d / pt3d, the match assembly of forward_finish) on the first `cap` match slots."""
        r = st["res"]
        B, M = r["i_ids"].shape
        if not (self.SPECULATE_SINGLE_PAIR and B == 1 and st["ffeat_of"] is None and not st["ret_feats"]):
            return
        cap = self._spec_cap(M)
        dev = st["dev"]
        i_c, j_c, c_c = r["i_ids"][0, :cap], r["j_ids"][0, :cap], r["mconf"][0, :cap]
        zkey = (str(dev), cap)
        zeros = self.__dict__.setdefault("_spec_zeros", {})
        if zkey not in zeros:
            zeros[zkey] = torch.zeros(cap, dtype=torch.int64, device=dev)
        b_c = zeros[zkey]
        expec = self._fine_stage(st["pt_cfeat"], st["im_ffeat"], b_c, i_c, j_c, r["count"], None)
        spec = dict(cap=cap, expec_f=expec, b_ids=b_c)
        if pt2d is not None and pt2d.dtype == torch.float32 and pt3d.dtype == torch.float32:
            # the seven indexing / elementwise launches of _assemble + the pred_mask compare as ONE (same expressions, same bits)
            mpt2d_c, mpt2d_f, mpt3d, mask = ops.assemble_matches(pt2d[0].contiguous(), pt3d[0].contiguous(), i_c, j_c, expec, c_c, self.win_sz, self.fine_ds)
            spec.update(mpt2d_c=mpt2d_c, mpt3d=mpt3d, mpt2d_f=mpt2d_f, pred_mask=mask)
        else:
            spec["pred_mask"] = c_c != 0
            if pt2d is not None:
                mpt2d_c, mpt3d = pt2d[0][i_c], pt3d[0][j_c]
                spec.update(mpt2d_c=mpt2d_c, mpt3d=mpt3d, mpt2d_f=mpt2d_c + expec[:, :2] * self.win_sz / 2 * self.fine_ds)
        st["spec"] = spec

    def _assemble(self, preds, pt2d, pt3d):
        b_ids, i_ids, j_ids = preds["match_ids"]
        mpt2d_c = pt2d[b_ids, i_ids]
        mpt3d = pt3d[b_ids, j_ids]
        mpt2d_f = mpt2d_c + preds["expec_f"][:, :2] * self.win_sz / 2 * self.fine_ds
        return b_ids, mpt2d_c, mpt2d_f, mpt3d

    def forward_multi_pair(self, data, mutual=False, match_thres=0.0):
        """Top-k reference frames (pt3d (B,k,N,3)).  The reference loops over the k frames and re-runs the WHOLE
        forward_match each time, image backbone and image self-attention included (c2f_trainer.py:385-399); the image
        side does not depend on the frame, so it is evaluated once here and the k point sets go through the point
        encoder, the cross attention and the matcher as ONE batch of B*k rows.  Outputs are concatenated in the
        reference's order (frame-major, then batch element, then image token)."""
        pt2d, pt3d, pt_feat, pt_mask = data["pt2d"], data["pt3d"], data["pt_feat"], data["pt_mask"]
        B, k, N, _ = pt3d.shape
        im_cfeat, im_ffeat = self.extract_im_feat(data["image"])
        pt_c = self.extract_pt_feat(pt_feat.reshape(B * k, N, -1), pt3d.reshape(B * k, N, 3))
        im_rep = im_cfeat.repeat_interleave(k, 0)                       # row b*k + j <-> (batch b, frame j)
        im_m = None if data["im_mask"] is None else data["im_mask"].repeat_interleave(k, 0)
        pt_m = None if pt_mask is None else pt_mask.reshape(B * k, N)
        try:
            preds = self._match_tokens(im_rep, im_ffeat, pt_c, im_m, pt_m, False, mutual, match_thres, ffeat_of=[r // k for r in range(B * k)])
        except StaleParameters:  # (ops.ParamGuard: parameters written through `.data`; once, on fresh copies)
            self.invalidate()
            return self.forward_multi_pair(data, mutual=mutual, match_thres=match_thres)
        rows, i_ids, j_ids = preds["match_ids"]
        b_ids, frame = rows // k, rows % k
        mpt2d_c = pt2d[b_ids, i_ids]
        mpt3d = pt3d.reshape(B * k, N, 3)[rows, j_ids]
        mpt2d_f = mpt2d_c + preds["expec_f"][:, :2] * self.win_sz / 2 * self.fine_ds
        order = torch.argsort(frame * (B * (i_ids.max() + 1 if len(i_ids) else 1)) + b_ids * (i_ids.max() + 1 if len(i_ids) else 1) + i_ids)
        data.update(dict(mpt2d_f=mpt2d_f[order], mpt2d_c=mpt2d_c[order], mpt3d=mpt3d[order], m_bids=b_ids[order], mconf=preds["mconf"][order]))

    # -- training --------------------------------------------------------------------------
    def forward_train(self, data, ret_feats=False, mutual=False, match_thres=0.0, alpha=0.25, gamma=2.0, train_percent=0.3, pad_gt=True):
        """forward(data, training=True) of the reference (c2f_trainer.py:429-488 with extract_mutual_matches' GT padding,
        extract_matches.py:38-56; pad_gt=False: the validation call, predicted matches only) through the autograd functions;
        additionally stores `coarse_loss`
        (compute_matching_loss, evaluated by the same kernels that hold the similarity matrix).  Must run inside
        `autograd.training()` (forward_with_metrics does that)."""
        pt2d, pt3d = data["pt2d"], data["pt3d"]
        preds = self._train_preds(data["image"], data["pt_feat"], pt3d, data["im_mask"], data["pt_mask"], data["conf_gt"], ret_feats=ret_feats,
                                  mutual=mutual, match_thres=match_thres, alpha=alpha, gamma=gamma, train_percent=train_percent, pad_gt=pad_gt)
        data.update(preds)
        b_ids, i_ids, j_ids = preds["match_ids"]
        _, mpt2d_c, mpt2d_f, mpt3d = self._assemble(preds, pt2d, pt3d)
        data.update(mpt2d_c_train=mpt2d_c, mpt3d_train=mpt3d, mpt2d_f_train=mpt2d_f)
        keep = preds["pred_mask"]
        kidx = torch.nonzero(keep).reshape(-1)  # ONE compaction (a scan, a count read-back) for the five selections below, not one each
        data.update(dict(m_bids=b_ids[kidx], mpt2d_c=mpt2d_c[kidx], mpt2d_f=mpt2d_f[kidx], mpt3d=mpt3d[kidx]))
        if "pt2d_proj" in data:
            gt = data["pt2d_proj"][b_ids, j_ids]
            data["mpt2d_f_gt_train"] = gt
            data["mpt2d_f_gt"] = gt[kidx]

    def _gt_ids(self, conf_gt):
        """torch.where(conf_gt) of the step's ground-truth mask -- a scan of B*M*N bytes plus a count read-back --, made once per tensor
        state and kept for the second caller of the same step."""
        key = (conf_gt.data_ptr(), conf_gt._version, tuple(conf_gt.shape))
        hit = self.__dict__.get("_gt_ids_cache")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_gt_ids_cache"] = (key, torch.where(conf_gt))
        return hit[1]

    def _train_preds(self, img, pt_feat, pt3d, im_mask, pt_mask, conf_gt, ret_feats=False, mutual=False, match_thres=0.0, alpha=0.25,
                     gamma=2.0, train_percent=0.3, pad_gt=True):
        """forward_match with ground truth (must run inside autograd.training()): encoders, cross attention, dual-softmax with
        the focal loss evaluated by the kernels that hold the similarity matrix (`coarse_loss`), GT-padded match sampling,
        fine stage.  Returns the reference's prediction dict."""
        import numpy as np

        im_cfeat, im_ffeat = self.extract_im_feat(img)
        pt_cfeat = self.extract_pt_feat(pt_feat, pt3d)
        im_cfeat, pt_cfeat = self.cross(im_cfeat, pt_cfeat)
        loss_c, conf, oi, oj, oc, cnt, im_n, pt_n = ag.coarse_match_loss(im_cfeat, pt_cfeat, self.temperature, self._match_scale(), im_mask,
                                                             pt_mask, conf_gt, self.temp_type, mutual, match_thres, alpha, gamma)
        B, M, N = conf.shape
        dev = conf.device
        counts = cnt.cpu().tolist()
        b_ids = torch.cat([torch.full((k,), b, device=dev, dtype=torch.int64) for b, k in enumerate(counts)])
        i_ids = torch.cat([oi[b, :k] for b, k in enumerate(counts)])
        j_ids = torch.cat([oj[b, :k] for b, k in enumerate(counts)])
        mconf = torch.cat([oc[b, :k] for b, k in enumerate(counts)])
        pred_num = int(b_ids.shape[0])
        # GT padding: a fixed number of training matches, `coarse_percent` of them predictions, the rest ground truth;
        # same numpy draws as the reference (np.random.choice on the global RNG)
        if pad_gt:
            total_pts = B * min(M, N)
            b_gt, i_gt, j_gt = self._gt_ids(conf_gt)
            train_num = int(total_pts * train_percent)
            pred_num = min(int(train_num * self.coarse_percent), pred_num)
            gt_num = train_num - pred_num
            pred_idx = torch.from_numpy(np.random.choice(len(b_ids), pred_num)).to(dev)
            gt_idx = torch.from_numpy(np.random.choice(len(b_gt), gt_num)).to(dev)
            b_ids = torch.cat([b_ids[pred_idx], b_gt[gt_idx]])
            i_ids = torch.cat([i_ids[pred_idx], i_gt[gt_idx]])
            j_ids = torch.cat([j_ids[pred_idx], j_gt[gt_idx]])
            mconf = torch.cat([mconf[pred_idx], torch.zeros(gt_num, device=dev)])
        # fine stage
        K = int(b_ids.shape[0])
        if K == 0:
            expec_f = torch.empty(0, 3, device=dev)
        else:
            pf = ag.linear(ag.linear(pt_cfeat, self.pt_ffeat_proj[0].weight, self.pt_ffeat_proj[0].bias), self.pt_ffeat_proj[1].weight,
                           self.pt_ffeat_proj[1].bias)
            pf = pf[b_ids, j_ids]
            win = self.fine_sa(ag.fine_windows(im_ffeat, b_ids, i_ids, self.win_sz, 4))
            expec_f = ag.fine_expectation(pf, win, self.win_sz)
        preds = dict(conf_matrix=conf, expec_f=expec_f, match_ids=(b_ids, i_ids, j_ids), mconf=mconf, pred_mask=mconf != 0, pred_num=pred_num,
                     coarse_loss=loss_c)
        if ret_feats:
            preds.update(im_cfeat=im_n, pt_cfeat=pt_n)
        return preds

    def forward_with_metrics(self, data, rthres=1, training=False, coarse_only=False, oracle=False):
        """Losses of one training / validation step (c2f_trainer.py:490-551): metrics["loss"] carries the autograd graph whose
        backward runs the HIP kernels.  The pose metrics of the reference (PnP on the matches; third-party solver) are not
        part of the loss and are not computed here."""
        if self.fine_loss not in ("match", "exp"):
            raise ValueError(self.fine_loss)
        metrics = {}
        with ag.training():
            self.forward_train(data, ret_feats=True, pad_gt=training)
            # feature distance of the ground-truth pairs (compute_feat_l2, utils/metrics.py:383-390; a logged diagnostic)
            im_n, pt_n = data.pop("im_cfeat"), data.pop("pt_cfeat")
            b_gt, i_gt, j_gt = self._gt_ids(data["conf_gt"])  # (the scan of the (B, M, N) mask made once per step, in _train_preds)
            dist_gt = (im_n[b_gt, i_gt] - pt_n[b_gt, j_gt]).norm(dim=-1)
            per_b = torch.zeros(im_n.shape[0], device=dist_gt.device).index_add_(0, b_gt, dist_gt)
            metrics["feat_l2"] = (per_b / torch.bincount(b_gt, minlength=im_n.shape[0])).mean()
            coarse_loss = data["coarse_loss"]
            metrics["coarse_loss"] = coarse_loss
            if len(data["match_ids"][1]) == 0 or coarse_only:
                metrics["loss"] = coarse_loss
                return metrics
            mpt2d_f_gt, mpt2d_f, mpt2d_c = data["mpt2d_f_gt_train"], data["mpt2d_f_train"], data["mpt2d_c_train"]
            coarse_dist = (mpt2d_f_gt - mpt2d_c).norm(dim=-1)
            coarse_pos = coarse_dist < self.coarse_dthres
            metrics["coarse_dist"] = coarse_dist.mean()
            metrics["coarse_pos_ratio"] = coarse_pos.float().mean() * 100
            # K-element arithmetic, plain tensor ops: the std of the window soft-max weighs the squared distance (detached)
            expec_f = data["expec_f"]
            inverse_std = 1.0 / torch.clamp(expec_f[:, 2], min=1e-10)
            weight = (inverse_std / torch.mean(inverse_std)).detach()
            if self.fine_loss == "match":  # compute_fine_match_loss_l2_std, utils/metrics.py:425-451: distance in image pixels
                mask = coarse_pos
                if mask.sum() == 0:
                    mask = mask.clone()
                    mask[0] = True
                    weight[0] = 0.0
                flow_l2 = ((mpt2d_f - mpt2d_f_gt) ** 2).sum(-1)
                fine_loss = (flow_l2 * weight * mask).mean()
            else:  # compute_fine_loss_l2_std, utils/metrics.py:393-422 (LoFTR's): distance in window units, correct cells only
                radius = self.fine_ds * self.win_sz // 2
                expec_gt = (mpt2d_f_gt - mpt2d_c) / radius
                correct = torch.linalg.norm(expec_gt, ord=float("inf"), dim=1) < 1
                if not correct.any():
                    correct = correct.clone()
                    correct[0] = True
                    weight[0] = 0.0
                flow_l2 = ((expec_gt[correct] - expec_f[correct, :2]) ** 2).sum(-1)
                fine_loss = (flow_l2 * weight[correct]).mean()
            metrics["fine_loss"] = fine_loss
            metrics["loss"] = coarse_loss + fine_loss
        return metrics

    def forward(self, data, training=False, ret_feats=False, mutual=False, match_thres=0.0):
        if training:
            with ag.training():
                return self.forward_train(data, ret_feats=ret_feats, mutual=mutual, match_thres=match_thres)
        pt3d, pt2d = data["pt3d"], data["pt2d"]
        if pt3d.dim() == 4:
            return self.forward_multi_pair(data, mutual=mutual, match_thres=match_thres)
        return self.forward_finish(self.forward_begin(data, ret_feats=ret_feats, mutual=mutual, match_thres=match_thres))

    def forward_image_side(self, img):
        """The image side of forward_match on its own -- backbone, tokens, sine PE, the self-attention block: (im_cfeat, im_ffeat).  It does
        not depend on the points, so a caller that still has to RENDER them can run it beside the render (NeRFMatchEvaluator does, on a
        compute-unit partition of its own) and hand the result to forward_begin(..., image_side=...).  Same values as the one-batch form
        (_shared_sa_batchable): every kernel of the block works per row / per sequence."""
        return self.extract_im_feat(img)

    def forward_begin(self, data, ret_feats=False, mutual=False, match_thres=0.0, image_side=None):
        """First half of forward() for a single-pair batch (see forward_match_begin); forward_finish(state) completes `data`."""
        if image_side is not None:
            im_cfeat, im_ffeat = image_side
            self._guard_early(data["pt3d"])
            pt_cfeat = self.extract_pt_feat(data["pt_feat"], data["pt3d"])
            st = self._match_tokens_begin(im_cfeat, im_ffeat, pt_cfeat, data["im_mask"], data["pt_mask"], ret_feats, mutual, match_thres)
        else:
            st = self.forward_match_begin(data["image"], data["pt_feat"], data["pt3d"], im_mask=data["im_mask"], pt_mask=data["pt_mask"],
                                          ret_feats=ret_feats, mutual=mutual, match_thres=match_thres)
        st["data"] = data
        st["kw"] = dict(ret_feats=ret_feats, mutual=mutual, match_thres=match_thres)
        st["all_pred"] = match_thres >= 0.0  # extracted matches have conf > match_thres >= 0: `mconf != 0` holds for all of them
        if data["pt2d"] is not None:
            self._speculate(st, data["pt2d"], data["pt3d"])
        return st

    def forward_finish(self, st):
        data = st["data"]
        pt3d, pt2d = data["pt3d"], data["pt2d"]
        try:
            preds = self.forward_match_finish(st)
        except StaleParameters:  # (ops.ParamGuard: parameters written through `.data`): the whole pass once more, image side included
            import warnings

            warnings.warn("nerfmatch_amd: matcher parameters were modified in place through `.data` (no version bump) after their packed copies "
                          "were made; the copies were rebuilt and this pass repeated -- call invalidate() after such writes to avoid the repetition")
            self.invalidate()
            return self.forward_finish(self.forward_begin(data, **st["kw"]))
        data.update(preds)
        spec, K = st.get("spec"), preds["pred_num"]
        if spec is not None and "mpt2d_f" in spec and K <= spec["cap"]:  # assembled before the read-back: slices only
            b_ids, mpt2d_c, mpt2d_f, mpt3d = preds["match_ids"][0], spec["mpt2d_c"][:K], spec["mpt2d_f"][:K], spec["mpt3d"][:K]
        else:
            b_ids, mpt2d_c, mpt2d_f, mpt3d = self._assemble(preds, pt2d, pt3d)
        data.update(mpt2d_c_train=mpt2d_c, mpt3d_train=mpt3d, mpt2d_f_train=mpt2d_f)
        # the reference keeps the rows with pred_mask = (mconf != 0), which drops only the GT-padded rows of training; at
        # inference every row passes, and boolean-mask indexing would cost a device synchronisation (torch.nonzero) that
        # drains whatever the caller queued behind this batch
        if st.get("all_pred", False):
            sel = lambda t: t
        else:
            keep = preds["pred_mask"]
            sel = lambda t: t[keep]
        data.update(dict(m_bids=sel(b_ids), mpt2d_c=sel(mpt2d_c), mpt2d_f=sel(mpt2d_f), mpt3d=sel(mpt3d)))
        if "pt2d_proj" in data:
            gt = data["pt2d_proj"][preds["match_ids"][0], preds["match_ids"][2]]
            data["mpt2d_f_gt_train"] = gt
            data["mpt2d_f_gt"] = sel(gt)


class NeRFMatcherCoarse(_MatcherBase):
    def __init__(self, config):
        super().__init__()
        self.coarse_ds = 8
        self.backbone = init_backbone(config.backbone, pretrained=getattr(config, "pretrained", False), downsample=self.coarse_ds)
        self._init_common(config)
        bd = self.backbone.feat_dim
        self.cfeat_proj = nn.Linear(bd, self.cfeat_dim, bias=True) if bd != self.cfeat_dim else None
        self.pt_feat_normalize = bool(getattr(config, "pt_feat_norm", False))
        self.keep_conf = True

    def extract_im_feat(self, img):
        cfeat = self.backbone(img)
        if isinstance(cfeat, (list, tuple)):
            cfeat = cfeat[0]
        return self.tokens_from_cfeat(cfeat)

    def forward_match(self, img, pt_feat, pt3d, im_mask=None, pt_mask=None, ret_feats=False, mutual=False, match_thres=0.0):
        def once():
            self._guard_early(pt3d)
            im = self.extract_im_feat(img)
            pt = self.extract_pt_feat(pt_feat, pt3d)
            im, pt = self.cross(im, pt)
            ids, mconf, conf, feats = self.coarse_match(im, pt, im_mask, pt_mask, mutual, match_thres, ret_feats, self.keep_conf)
            preds = dict(conf_matrix=conf, match_ids=ids, mconf=mconf, pred_num=ids[0].shape[0])
            if ret_feats:
                preds.update(im_cfeat=feats[0], pt_cfeat=feats[1])
            return preds

        return self._retry_if_stale(once)  # (ops.ParamGuard: parameters written through `.data` -> once more on fresh copies)

    def forward_with_metrics(self, data, rthres=1, training=False, coarse_only=False):
        """Loss of one training / validation step of the coarse-only model (nerfmatch_coarse_trainer.py:365-387): the focal
        loss WITHOUT the confidence clamp (`clamp=False`, :380); the pose metrics (PnP, third party) are not computed."""
        with ag.training():
            im = self.extract_im_feat(data["image"])
            pt = self.extract_pt_feat(data["pt_feat"], data["pt3d"])
            im, pt = self.cross(im, pt)
            loss, conf, oi, oj, oc, cnt, im_n, pt_n = ag.coarse_match_loss(im, pt, self.temperature, self._match_scale(), data["im_mask"],
                                                                           data["pt_mask"], data["conf_gt"], self.temp_type, False, 0.0,
                                                                           clamp=False)
        counts = cnt.cpu().tolist()
        dev = conf.device
        ids = (torch.cat([torch.full((k,), b, device=dev, dtype=torch.int64) for b, k in enumerate(counts)]),
               torch.cat([oi[b, :k] for b, k in enumerate(counts)]), torch.cat([oj[b, :k] for b, k in enumerate(counts)]))
        data.update(conf_matrix=conf, match_ids=ids, mconf=torch.cat([oc[b, :k] for b, k in enumerate(counts)]), pred_num=ids[0].shape[0])
        return dict(coarse_loss=loss, loss=loss)

    def forward_multi_pair(self, data, mutual=False, match_thres=0.0):
        b, i, j, c = [], [], [], []
        for ipt3d, ipt_feat, ipt_mask in zip(data["pt3d"].permute(1, 0, 2, 3), data["pt_feat"].permute(1, 0, 2, 3),
                                             data["pt_mask"].permute(1, 0, 2)):
            p = self.forward_match(data["image"], ipt_feat, ipt3d, im_mask=data["im_mask"], pt_mask=ipt_mask, mutual=mutual,
                                   match_thres=match_thres)
            b.append(p["match_ids"][0]); i.append(p["match_ids"][1]); j.append(p["match_ids"][2]); c.append(p["mconf"])
        data.update(dict(match_ids=(torch.cat(b), torch.cat(i), torch.cat(j)), mconf=torch.cat(c)))
        return data

    def forward(self, data, ret_feats=False, mutual=False, match_thres=0.0):
        if data["pt3d"].dim() == 4:
            return self.forward_multi_pair(data, mutual=mutual, match_thres=match_thres)
        data.update(self.forward_match(data["image"], data["pt_feat"], data["pt3d"], im_mask=data["im_mask"], pt_mask=data["pt_mask"],
                                       ret_feats=ret_feats, mutual=mutual, match_thres=match_thres))
        return data
