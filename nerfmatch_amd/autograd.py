"""Training side of the matcher head: torch.autograd.Function wrappers whose forward AND
backward are the HIP kernels of csrc/ (torch's autograd engine only orders the calls and sums fan-out gradients).

The reference trains through eager autograd (NeRFMatcherMS.forward_with_metrics, nerfmatch_c2f_trainer.py:490-551;
compute_matching_loss, utils/metrics.py:372-380).  The modules of nerfmatch_amd switch to these functions inside a
`with autograd.training():` block (entered by forward_with_metrics); outside it they run the fused inference kernels and
build no graph.  No CPU / eager fallback: every op below raises if the HIP library is missing.
"""
import contextlib
import ctypes as C

import torch
from torch.autograd import Function

from . import _lib, ops
from ._lib import check, dptr, lib, stream

_TRAINING = False


@contextlib.contextmanager
def training(flag=True):
    global _TRAINING
    prev, _TRAINING = _TRAINING, bool(flag)
    try:
        yield
    finally:
        _TRAINING = prev


def is_training():
    return _TRAINING and torch.is_grad_enabled()


class _Linear(Function):
    """y = x @ w.T (+ bias) (+ residual)   -- nm_linear forward, nm_linear (dx) / nm_linear_wgrad (dw) / nm_col_sum (db) backward."""

    @staticmethod
    def forward(ctx, x, w, bias, residual):
        K, N = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, K).contiguous()
        y = ops.linear(x2, w.detach(), None if bias is None else bias.detach(), residual=None if residual is None else residual.detach())
        ctx.save_for_backward(x2, w)
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        N = w.shape[0]
        dy2 = dy.reshape(-1, N).contiguous()
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_t(dy2, w).reshape(ctx.xshape)  # dy @ W on the blob of W^T packed straight from W
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dw, db = ops.linear_wgrad_bias(dy2, x2)
        elif ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(dy2, x2)
        elif ctx.needs_input_grad[2]:
            db = ops.col_sum(dy2)
        if ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres


def linear(x, w, bias=None, residual=None):
    return _Linear.apply(x, w, bias, residual)


class _LinearRelu(Function):
    """y = relu(x @ w.T + bias): nm_linear with the fused activation forward; backward through nm_relu_bwd and the linear layer's kernels."""

    @staticmethod
    def forward(ctx, x, w, bias):
        K, N = x.shape[-1], w.shape[0]
        x2 = x.reshape(-1, K).contiguous()
        y = ops.linear(x2, w.detach(), None if bias is None else bias.detach(), act=_lib.NM_ACT_RELU)
        ctx.save_for_backward(x2, w, y)
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        g = ops.relu_bwd(y, dy.reshape(-1, w.shape[0]))
        dx = ops.linear_t(g, w).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            dw, db = ops.linear_wgrad_bias(g, x2)
        else:
            dw = ops.linear_wgrad(g, x2) if ctx.needs_input_grad[1] else None
            db = ops.col_sum(g) if ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear_relu(x, w, bias=None):
    return _LinearRelu.apply(x, w, bias)


class _Gelu(Function):
    @staticmethod
    def forward(ctx, u):
        ctx.save_for_backward(u)
        return ops.gelu(u)

    @staticmethod
    def backward(ctx, dh):
        (u,) = ctx.saved_tensors
        return ops.gelu_bwd(u, dh)


def gelu(u):
    return _Gelu.apply(u)


class _EncoderTailFrozen(Function):
    """y = xh + FFN(LN2(xh + att @ Wo.T)) with FROZEN parameters (round 6): the forward is the inference path's fused tail that also keeps the
    two intermediates the backward needs -- the LayerNorm's input and the GELU's input -- (ops.encoder_tail_save: one launch, five as separate
    kernels), the backward is ONE kernel (ops.encoder_tail_bwd) instead of three GEMM launches, gelu_bwd, layernorm_bwd and the adds autograd
    inserts where xh fans out.  Only att and xh receive gradients."""

    @staticmethod
    def forward(ctx, att, xh, w_out, norm2, ffn0, ffn2):
        y, a, u = ops.encoder_tail_save(att, xh, w_out.detach(), norm2, ffn0, ffn2)  # one launch (five as separate kernels)
        ctx.save_for_backward(a, u)
        ctx.mods = (w_out, norm2, ffn0, ffn2)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, u = ctx.saved_tensors
        w_out, norm2, ffn0, ffn2 = ctx.mods
        d_att, d_xh = ops.encoder_tail_bwd(dy.contiguous(), a, u, w_out.detach(), norm2, ffn0, ffn2)
        return d_att, d_xh, None, None, None, None


def encoder_tail_frozen(att, xh, w_out, norm2, ffn0, ffn2):
    return _EncoderTailFrozen.apply(att, xh, w_out, norm2, ffn0, ffn2)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return ops.layernorm(x, gamma.detach(), beta.detach(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, gamma.detach(), dy, ctx.eps, param_grads=ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, dg, db, None


def layernorm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        o = ops.attention(q, k, v, heads, scale)
        ctx.save_for_backward(q, k, v, o)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, k, v, o = ctx.saved_tensors
        dq, dk, dv = ops.attention_bwd(q, k, v, o, d_o, ctx.heads, ctx.scale)
        return dq, dk, dv, None, None


def attention(q, k, v, heads, scale):
    return _Attention.apply(q, k, v, heads, scale)


class _AttentionSelfFused(Function):
    """Self attention reading q | k | v as the three column blocks of ONE fused projection buffer (B*L, 3*dim)."""

    @staticmethod
    def forward(ctx, qkv, B, L, heads, scale):
        dim = qkv.shape[1] // 3
        qkv = qkv.contiguous()
        o, nlse = ops.attention_fused(qkv, (0, dim), (dim, 2 * dim), (2 * dim, 3 * dim), B, L, L, heads, scale, want_lse=True)
        ctx.save_for_backward(qkv, o)
        ctx.meta, ctx.nlse = (B, L, heads, scale, dim), nlse  # (the forward kernel's log-sum-exp: the backward makes one pass over the keys less)
        return o

    @staticmethod
    def backward(ctx, d_o):
        qkv, o = ctx.saved_tensors
        B, L, heads, scale, dim = ctx.meta
        dqkv, _ = ops.attention_bwd_fused(qkv, 0, qkv, dim, 2 * dim, o, d_o, B, L, L, heads, scale, nlse=ctx.nlse)
        return dqkv, None, None, None, None


class _AttentionCrossFused(Function):
    """Cross attention: q (B*L, dim) and the fused [k | v] projection buffer (B*S, 2*dim)."""

    @staticmethod
    def forward(ctx, q, kv, B, L, S, heads, scale):
        dim = q.shape[1]
        q, kv = q.contiguous(), kv.contiguous()
        o, nlse = ops.attention_fused(q, (0, dim), (0, dim), (dim, 2 * dim), B, L, S, heads, scale, kv=kv, want_lse=True)
        ctx.save_for_backward(q, kv, o)
        ctx.meta, ctx.nlse = (B, L, S, heads, scale, dim), nlse
        return o

    @staticmethod
    def backward(ctx, d_o):
        q, kv, o = ctx.saved_tensors
        B, L, S, heads, scale, dim = ctx.meta
        dq, dkv = ops.attention_bwd_fused(q, 0, kv, 0, dim, o, d_o, B, L, S, heads, scale, nlse=ctx.nlse)
        return dq, dkv, None, None, None, None, None


def attention_self_fused(qkv, B, L, heads, scale):
    return _AttentionSelfFused.apply(qkv, B, L, heads, scale)


def attention_cross_fused(q, kv, B, L, S, heads, scale):
    return _AttentionCrossFused.apply(q, kv, B, L, S, heads, scale)


class _TokensFromMap(Function):
    """(B,C,h,w) feature map -> (B,h*w,C) tokens (+ sine PE table); backward is the inverse permutation."""

    @staticmethod
    def forward(ctx, x, pe_table):
        ctx.shape = x.shape
        return ops.nchw_to_tokens(x, pe_table)

    @staticmethod
    def backward(ctx, dy):
        B, Cc, h, w = ctx.shape
        return dy.reshape(B, h, w, Cc).permute(0, 3, 1, 2).contiguous(), None


def tokens_from_map(x, pe_table=None):
    return _TokensFromMap.apply(x, pe_table)


class _CatFourier(Function):
    """[feat | fourier(pt3d)] zero-padded to a multiple of 8 columns.  `pt3d` receives a gradient only when it asks for one
    (the iNeRF matching term; training data never does)."""

    @staticmethod
    def forward(ctx, feat, pt3d, num_freqs):
        ctx.C, ctx.num_freqs = feat.shape[-1], num_freqs
        pt3d = pt3d.contiguous()
        ctx.save_for_backward(pt3d)
        return ops.cat_fourier(feat.contiguous(), pt3d, num_freqs)

    @staticmethod
    def backward(ctx, dy):
        g_pt = ops.cat_fourier_bwd(dy, ctx.saved_tensors[0], ctx.C, ctx.num_freqs) if ctx.needs_input_grad[1] else None
        return dy[:, : ctx.C].contiguous() if ctx.needs_input_grad[0] else None, g_pt, None


def cat_fourier(feat, pt3d, num_freqs=15):
    return _CatFourier.apply(feat, pt3d, num_freqs)


class _CoarseMatchLoss(Function):
    """Dual-softmax matching + focal loss over the batch (coarse_matching c2f_trainer.py:289-300 + compute_matching_loss
    utils/metrics.py:372-380).  Returns (loss, conf_matrix, i_ids, j_ids, mconf, counts, im_norm, pt_norm); only `loss` is
    differentiable."""

    @staticmethod
    def forward(ctx, im, pt, temperature, scale, im_mask, pt_mask, conf_gt, temp_type, mutual, threshold, alpha, gamma, clamp, loss_only=False):
        B, M, Cc = im.shape
        N = pt.shape[1]
        dev = im.device
        L = lib()
        scale = float(scale)  # host value of the temperature (T or 1/T), cached by the model
        gt = conf_gt.to(torch.uint8).contiguous()
        acc = torch.zeros(4, device=dev, dtype=torch.float64)
        check(L.nm_focal_count(dptr(gt, torch.uint8), gt.numel(), dptr(acc, torch.float64), stream()), "nm_focal_count")
        need = L.nm_match_workspace_bytes(M, N, Cc)
        flags = _lib.NM_MATCH_BF16X3 if ops.MATCH_PRECISION == "bf16x3" else 0
        if loss_only:  # (similarity + statistics only: no confidence matrix, no selection -- the loss reads neither)
            flags |= _lib.NM_MATCH_STATS_ONLY
        im_c, pt_c = im.detach().contiguous(), pt.detach().contiguous()
        im_m = None if im_mask is None else im_mask.to(torch.uint8).contiguous()
        pt_m = None if pt_mask is None else pt_mask.to(torch.uint8).contiguous()
        Mo = 0 if loss_only else M
        conf = torch.empty(B, Mo, N, device=dev, dtype=torch.float32)
        imn, ptn = torch.empty_like(im_c), torch.empty_like(pt_c)
        oi = torch.empty(B, Mo, device=dev, dtype=torch.int64)
        oj = torch.empty(B, Mo, device=dev, dtype=torch.int64)
        oc = torch.empty(B, Mo, device=dev, dtype=torch.float32)
        cnt = torch.zeros(B, device=dev, dtype=torch.int32)
        row_t = torch.empty(B, M, device=dev, dtype=torch.float32)
        col_t = torch.empty(B, N, device=dev, dtype=torch.float32)
        wss = []
        for b in range(B):
            ws = torch.empty(need, device=dev, dtype=torch.uint8)  # kept for the backward pass (similarity + statistics)
            wss.append(ws)
            mi = None if im_m is None else im_m[b]
            mp = None if pt_m is None else pt_m[b]
            check(L.nm_dual_softmax_match_ex(dptr(im_c[b]), dptr(pt_c[b]), M, N, Cc, scale, dptr(mi, torch.uint8), dptr(mp, torch.uint8),
                                             float(threshold), int(bool(mutual)), flags, None if loss_only else dptr(conf[b]), dptr(imn[b]), dptr(ptn[b]),
                                             None if loss_only else dptr(oi[b], torch.int64), None if loss_only else dptr(oj[b], torch.int64),
                                             None if loss_only else dptr(oc[b]), C.c_void_p(cnt.data_ptr() + 4 * b), dptr(ws, torch.uint8), need, stream()),
                  "nm_dual_softmax_match_ex")
            check(L.nm_match_focal_loss(dptr(gt[b], torch.uint8), M, N, Cc, float(alpha), float(gamma), int(bool(clamp)), dptr(ws, torch.uint8), need,
                                        dptr(acc, torch.float64), dptr(row_t[b]), dptr(col_t[b]), stream()), "nm_match_focal_loss")
        loss = (acc[0] / acc[2] + acc[1] / acc[3]).to(torch.float32)
        ctx.save_for_backward(im_c, pt_c, imn, ptn, gt, acc, row_t, col_t)
        ctx.wss, ctx.masks, ctx.scale, ctx.temp_type = wss, (im_m, pt_m), scale, temp_type
        ctx.alpha, ctx.gamma, ctx.clamp = float(alpha), float(gamma), int(bool(clamp))
        imn_out, ptn_out = imn.clone(), ptn.clone()
        ctx.mark_non_differentiable(conf, oi, oj, oc, cnt, imn_out, ptn_out)
        return loss, conf, oi, oj, oc, cnt, imn_out, ptn_out

    @staticmethod
    def backward(ctx, g_loss, *unused):
        im_c, pt_c, imn, ptn, gt, acc, row_t, col_t = ctx.saved_tensors
        B, M, Cc = im_c.shape
        N = pt_c.shape[1]
        if N % 8 != 0:
            raise _lib.NerfmatchAmdError("the matching-loss backward needs a multiple of 8 point tokens")
        dev = im_c.device
        L = lib()
        im_m, pt_m = ctx.masks
        g = g_loss.detach().to(torch.float32).reshape(1).contiguous()
        dscale = torch.zeros(1, device=dev, dtype=torch.float64)
        d_im, d_pt = torch.empty_like(im_c), torch.empty_like(pt_c)
        ddot = torch.empty(M, N, device=dev, dtype=torch.float32)
        for b in range(B):
            ws = ctx.wss[b]
            mi = None if im_m is None else im_m[b]
            mp = None if pt_m is None else pt_m[b]
            check(L.nm_match_focal_loss_bwd(dptr(gt[b], torch.uint8), dptr(mi, torch.uint8), dptr(mp, torch.uint8), M, N, Cc, ctx.alpha,
                                            ctx.gamma, ctx.clamp, ctx.scale, dptr(g), dptr(ws, torch.uint8), ws.numel(), dptr(acc, torch.float64),
                                            dptr(row_t[b]), dptr(col_t[b]), dptr(ddot), dptr(dscale, torch.float64), stream()),
                  "nm_match_focal_loss_bwd")
            d_imn = ops.linear(ddot, ptn[b].t().contiguous())  # (M,N) @ (N,C)
            d_ptn = ops.linear_wgrad(ddot, imn[b])              # (M,N)^T @ (M,C)
            d_im[b] = ops.l2norm_bwd(im_c[b], d_imn)
            d_pt[b] = ops.l2norm_bwd(pt_c[b], d_ptn)
        ctx.wss = None
        d_temp = None
        if ctx.needs_input_grad[2]:
            # scale = T ("mul") or 1 / T ("div")
            d_temp = dscale.to(torch.float32).reshape(()) if ctx.temp_type == "mul" else (-(ctx.scale**2) * dscale).to(torch.float32).reshape(())
        return d_im, d_pt, d_temp, None, None, None, None, None, None, None, None, None, None, None


def coarse_match_loss(im, pt, temperature, scale, im_mask, pt_mask, conf_gt, temp_type="mul", mutual=False, threshold=0.0, alpha=0.25,
                      gamma=2.0, clamp=True, loss_only=False):
    """`scale` is the host value multiplying the cosine similarity (T for temp_type "mul", 1/T for "div").  loss_only: the confidence matrix
    and the match lists come back empty (the matching term of the pose refinement reads the loss alone)."""
    return _CoarseMatchLoss.apply(im, pt, temperature, scale, im_mask, pt_mask, conf_gt, temp_type, mutual, threshold, alpha, gamma, clamp,
                                  loss_only)


class _FineWindows(Function):
    """5x5 windows of the fine feature maps (B,C,Hf,Wf) at coarse cells (b_ids, i_ids), any order -> (K, 25, C)."""

    @staticmethod
    def forward(ctx, ffeat, b_ids, i_ids, win, stride):
        B = ffeat.shape[0]
        K = b_ids.shape[0]
        out = torch.empty(K, win * win, ffeat.shape[1], device=ffeat.device, dtype=torch.float32)
        groups = []
        for b in range(B):
            sel = (b_ids == b).nonzero().flatten()
            if sel.numel() == 0:
                continue
            ib = i_ids[sel].contiguous()
            cnt = torch.tensor([sel.numel()], device=ffeat.device, dtype=torch.int32)
            out[sel] = ops.fine_windows(ffeat[b].detach().contiguous(), ib, cnt, win, stride)
            groups.append((b, sel, ib, cnt))
        ctx.groups, ctx.shape, ctx.win, ctx.stride = groups, ffeat.shape, win, stride
        return out

    @staticmethod
    def backward(ctx, dwin):
        B, Cc, Hf, Wf = ctx.shape
        d = torch.zeros(B, Cc, Hf, Wf, device=dwin.device, dtype=torch.float32)
        for b, sel, ib, cnt in ctx.groups:
            ops.fine_windows_bwd(dwin[sel], (Cc, Hf, Wf), ib, cnt, ctx.win, ctx.stride, out=d[b])
        return d, None, None, None, None


def fine_windows(ffeat, b_ids, i_ids, win=5, stride=4):
    return _FineWindows.apply(ffeat, b_ids, i_ids, win, stride)


class _FineExpectation(Function):
    @staticmethod
    def forward(ctx, pt_f, win_f, win):
        K = pt_f.shape[0]
        cnt = torch.tensor([K], device=pt_f.device, dtype=torch.int32)
        pt_f, win_f = pt_f.contiguous(), win_f.contiguous()
        ctx.save_for_backward(pt_f, win_f, cnt)
        ctx.win = win
        return ops.fine_expectation(pt_f, win_f, cnt, win)

    @staticmethod
    def backward(ctx, d_expec):
        pt_f, win_f, cnt = ctx.saved_tensors
        d_pt, d_win = ops.fine_expectation_bwd(pt_f, win_f, d_expec, cnt, ctx.win)
        return d_pt, d_win, None


def fine_expectation(pt_f, win_f, win=5):
    return _FineExpectation.apply(pt_f, win_f, win)
