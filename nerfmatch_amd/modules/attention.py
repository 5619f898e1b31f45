"""Transformer encoder pieces with the reference's module tree (nerfmatch/modules/attention.py:84-285), so the
state-dict keys `...attention.proj_{q,k,v}.weight`, `...attention.proj_out.0.weight`, `...norm1.{0,1}.*`,
`...feedforward.layers.{0,2}.*`, `...norm2.*` are identical.  The forward passes call the HIP kernels
(nm_layernorm, nm_linear on the fp32 matrix cores, nm_attention flash-style) through nerfmatch_amd.ops."""
import torch
from torch import nn

from .. import autograd as ag
from .. import ops
from .._lib import NM_ACT_GELU, NM_ACT_RELU


class FullAttention(nn.Module):
    def __init__(self, head_dim):
        super().__init__()
        self.temperature = head_dim**0.5

    def scale(self):
        return 1.0 / self.temperature


class LocalitySelfAttention(nn.Module):
    """Learnable temperature; the diagonal mask of the reference is computed but never applied
    (attention.py:75-79), so the arithmetic is plain softmax attention with scale = exp(self.scale)."""

    def __init__(self, head_dim):
        super().__init__()
        self.scale_param_name = "scale"
        self.scale = nn.Parameter(torch.log(torch.tensor(head_dim**-0.5)))

    def scale_value(self):
        """exp(scale) as a host float, read back once per parameter state (a read-back is a full synchronisation; writes through `.data`
        are caught by ops.ParamGuard like every other derived copy)."""
        p = self.scale
        key = (p.data_ptr(), p._version)
        hit = self.__dict__.get("_scale_host")
        if hit is None or hit[0] != key:
            hit = self.__dict__["_scale_host"] = (key, float(p.detach().exp()))
        return hit[1]


class MultiHeadAttention(nn.Module):
    def __init__(self, model_dim, context_dim=None, head_num=8, head_dim=64, att_type="full", dropout=0.0):
        super().__init__()
        if dropout > 0:
            raise NotImplementedError("dropout is a training-time option")
        self.head_dim, self.head_num = head_dim, head_num
        inner = head_dim * head_num
        context_dim = context_dim or model_dim
        self.proj_q = nn.Linear(model_dim, inner, bias=False)
        self.proj_k = nn.Linear(context_dim, inner, bias=False)
        self.proj_v = nn.Linear(context_dim, inner, bias=False)
        if att_type == "full":
            self.attend = FullAttention(head_dim)
        elif att_type == "lsa":
            self.attend = LocalitySelfAttention(head_dim)
        else:
            raise TypeError(f"Unexpected att_type={att_type}")
        self.att_type = att_type
        self.proj_out = nn.Sequential(nn.Linear(inner, model_dim, bias=False))

    def _fused_weight(self, names):
        """Concatenation of projection weights (rows), cached until one of them changes."""
        ws = [getattr(self, n).weight for n in names]
        key = tuple((w.data_ptr(), w._version) for w in ws)
        cache = self.__dict__.setdefault("_fused", {})
        hit = cache.get(names)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([w.detach() for w in ws], 0).contiguous())
            cache[names] = hit
        return hit[1]

    def forward(self, query, key, value, residual=None, project_out=True):
        """project_out=False returns the attention output BEFORE proj_out (the fused encoder tail applies it).
        q/k/v projections are ONE GEMM when the inputs coincide (self attention: [q|k|v], cross attention: [k|v]); on the
        split-bf16 path that GEMM writes the keys / values directly as the attention kernel's pre-split MFMA operands
        (ops.attention_projected), otherwise the attention kernel reads the column slices in place (nm_attention_ld)."""
        if ag.is_training():
            return self._forward_train(query, key, value, residual, project_out)
        scale = self.attend.scale() if self.att_type == "full" else self.attend.scale_value()
        B, L, _ = query.shape
        S = key.shape[1]
        inner = self.head_dim * self.head_num
        fuse = key is value and self.att_type == "full" and ops.projected_attention_supported(key.shape[-1], self.head_num, self.head_dim, L, S)
        if fuse and query is key:
            att = ops.attention_projected(query.reshape(B * L, -1), None, None, self._fused_weight(("proj_q", "proj_k", "proj_v")), B, L, S,
                                          self.head_num, scale)
        elif fuse:
            att = ops.attention_projected(query.reshape(B * L, -1), self.proj_q.weight, key.reshape(B * S, -1),
                                          self._fused_weight(("proj_k", "proj_v")), B, L, S, self.head_num, scale)
        elif key is value and query is key:
            qkv = ops.linear(query.reshape(B * L, -1), self._fused_weight(("proj_q", "proj_k", "proj_v")))
            att = ops.attention_fused(qkv, (0, inner), (inner, 2 * inner), (2 * inner, 3 * inner), B, L, S, self.head_num, scale)
        elif key is value:
            q = ops.linear(query.reshape(B * L, -1), self.proj_q.weight)
            kv = ops.linear(key.reshape(B * S, -1), self._fused_weight(("proj_k", "proj_v")))
            att = ops.attention_fused(q, (0, inner), (0, inner), (inner, 2 * inner), B, L, S, self.head_num, scale, kv=kv)
        else:
            q = ops.linear(query, self.proj_q.weight)
            k = ops.linear(key, self.proj_k.weight)
            v = ops.linear(value, self.proj_v.weight)
            att = ops.attention(q, k, v, self.head_num, scale)
        if not project_out:
            return att
        return ops.linear(att, self.proj_out[0].weight, residual=residual)


    def _forward_train(self, query, key, value, residual, project_out=True):
        """Same arithmetic through the autograd functions (att_type "lsa": the learnable log-scale enters the graph through the queries)."""
        B, L, _ = query.shape
        S = key.shape[1]
        lsa = self.att_type == "lsa"
        scale = self.attend.scale_value() if lsa else self.attend.scale()
        inner = self.head_dim * self.head_num

        def learnable_scale(q):
            """LocalitySelfAttention (attention.py:60-81): the scores are q.k exp(p) with p a PARAMETER.  The kernels take the scale as a host
            number; the graph gets p through the queries -- q * (exp(p) / host value of exp(p)) is q times exactly 1.0 with d/dp = sum q . dq,
            which is d loss / d p of the scaled scores."""
            return q * (self.attend.scale.exp() / scale) if lsa else q
        # fused projections as in the inference path: one GEMM forward, one for dx and one weight-gradient GEMM backward
        # (torch.cat is differentiable plumbing: its backward hands each projection its row block of the fused gradient)
        # Frozen parameters (the matching term of the iNeRF refinement differentiates through the matcher w.r.t. its INPUTS, five times per
        # query): the stacked weights come from the inference path's cache -- no cat launch per call, and the packed / transposed copies
        # the GEMMs need (ops._linear_blob, ops.transposed: keyed on the tensor) are found again instead of being made anew every call.
        frozen = not (self.proj_q.weight.requires_grad or self.proj_k.weight.requires_grad or self.proj_v.weight.requires_grad)
        stack = (lambda names: self._fused_weight(names)) if frozen else (lambda names: torch.cat([getattr(self, n).weight for n in names], 0))
        if key is value and query is key:
            w = stack(("proj_q", "proj_k", "proj_v"))
            qkv = ag.linear(query.reshape(B * L, -1), w)
            if lsa:
                qkv = torch.cat([learnable_scale(qkv[:, :inner]), qkv[:, inner:]], 1)
            att = ag.attention_self_fused(qkv, B, L, self.head_num, scale)
        elif key is value:
            q = learnable_scale(ag.linear(query.reshape(B * L, -1), self.proj_q.weight))
            kv = ag.linear(key.reshape(B * S, -1), stack(("proj_k", "proj_v")))
            att = ag.attention_cross_fused(q, kv, B, L, S, self.head_num, scale)
        else:
            att = ag.attention(learnable_scale(ag.linear(query, self.proj_q.weight)), ag.linear(key, self.proj_k.weight),
                               ag.linear(value, self.proj_v.weight), self.head_num, scale)
        if not project_out:
            return att.reshape(B, L, inner)
        return ag.linear(att, self.proj_out[0].weight, residual=residual)


_ACTS = {"relu": NM_ACT_RELU, "gelu": NM_ACT_GELU}
_ACT_MODULES = {"relu": nn.ReLU, "gelu": nn.GELU}


class FeedForwardNetwork(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_dim=None, act_fn="relu", dropout=0.0, bias=True):
        super().__init__()
        if act_fn not in _ACTS or dropout:
            raise NotImplementedError(f"act_fn={act_fn} dropout={dropout}")
        hidden_dim = hidden_dim or in_dim
        self.act = _ACTS[act_fn]
        # index 1 is the (parameter-free) activation, so the Linear layers keep the reference's keys layers.0 / layers.2
        self.layers = nn.Sequential(nn.Linear(in_dim, hidden_dim, bias=bias), _ACT_MODULES[act_fn](), nn.Linear(hidden_dim, out_dim, bias=bias))

    def forward(self, x, residual=None):
        if ag.is_training():
            if self.act == NM_ACT_GELU:
                h = ag.gelu(ag.linear(x, self.layers[0].weight, self.layers[0].bias))
            else:  # nn.ReLU: fused into the first GEMM forward, nm_relu_bwd backward
                h = ag.linear_relu(x, self.layers[0].weight, self.layers[0].bias)
            return ag.linear(h, self.layers[2].weight, self.layers[2].bias, residual=residual)
        h = ops.linear(x, self.layers[0].weight, self.layers[0].bias, act=self.act)
        return ops.linear(h, self.layers[2].weight, self.layers[2].bias, residual=residual)


class GenericEncoderLayer(nn.Module):
    def __init__(self, model_dim=512, context_dim=None, head_num=8, head_dim=64, norm_type="pre", act_fn="gelu",
                 att_type="full", att_mode="self", dropout=0.0):
        super().__init__()
        assert not (att_type == "lsa" and att_mode == "cross"), "LocalSelfAttention is not suitable for cross attention!"
        self.norm_type, self.att_mode = norm_type, att_mode
        context_dim = context_dim or model_dim
        self.attention = MultiHeadAttention(model_dim, context_dim=context_dim, head_num=head_num, head_dim=head_dim,
                                            att_type=att_type, dropout=dropout)
        norms = [nn.LayerNorm(model_dim)]
        if norm_type == "pre" and att_mode == "cross":  # (post-norm: one LayerNorm also in cross mode, reference attention.py:195-198)
            norms.append(nn.LayerNorm(context_dim))
        self.norm1 = nn.Sequential(*norms)
        self.feedforward = FeedForwardNetwork(model_dim, model_dim, act_fn=act_fn, dropout=dropout)
        self.norm2 = nn.LayerNorm(model_dim)

    def forward(self, x, context=None):
        """y = xh + FFN(LN2(xh + MHA(xh, ch))), xh = LN1[0](x): both residuals add onto the NORMALISED input
        (reference attention.py:229-240)."""
        if self.att_mode == "self":
            assert context is None, "self attention does not expect extra context"
        ln = ag.layernorm if ag.is_training() else ops.layernorm
        if self.norm_type != "pre":
            # post-norm (reference forward_post_norm, attention.py:209-221; no shipped yaml selects it: the four separate launches, no fused
            # tail): a = LN1(x + MHA(x, ctx)); y = LN2(x + FFN(a)) -- the second residual is again the RAW input
            ctx = x if context is None else context
            n1 = self.norm1[0]
            a = ln(self.attention(x, ctx, ctx, residual=x), n1.weight, n1.bias, n1.eps)
            return ln(self.feedforward(a, residual=x), self.norm2.weight, self.norm2.bias, self.norm2.eps)
        n0 = self.norm1[0]
        if self.att_mode == "cross" and not ag.is_training():
            xh, ch = ops.layernorm_pair(x, n0, context, self.norm1[1])  # one launch, the same bits
        else:
            xh = ln(x, n0.weight, n0.bias, n0.eps)
            if self.att_mode == "cross":
                n1 = self.norm1[1]
                ch = ln(context, n1.weight, n1.bias, n1.eps)
            else:
                ch = xh
        ff = self.feedforward
        if (not ag.is_training() and ff.layers[0].bias is not None and ff.layers[2].bias is not None
                and ops.encoder_tail_supported(xh.shape[-1], self.attention.head_dim * self.attention.head_num, ff.layers[0].out_features, ff.act)):
            # proj_out + residual + norm2 + feed-forward + residual in one launch: three tensors through HBM instead of ten
            att = self.attention(xh, ch, ch, project_out=False)
            return ops.encoder_tail(att, xh, self.attention.proj_out[0].weight, self.norm2, ff.layers[0], ff.layers[2])
        if (ag.is_training() and ops.ENCODER_TAIL_BWD_FUSED and ff.layers[0].bias is not None and ff.layers[2].bias is not None
                and ops.encoder_tail_supported(xh.shape[-1], self.attention.head_dim * self.attention.head_num, ff.layers[0].out_features, ff.act)
                and not any(p.requires_grad for p in (self.attention.proj_out[0].weight, self.norm2.weight, self.norm2.bias, ff.layers[0].weight,
                                                      ff.layers[0].bias, ff.layers[2].weight, ff.layers[2].bias))):
            # frozen parameters (the iNeRF refinement's matching term): the tail's backward is ONE kernel (csrc/encoder_tail_bwd.hip)
            att = self.attention(xh, ch, ch, project_out=False)
            return ag.encoder_tail_frozen(att, xh, self.attention.proj_out[0].weight, self.norm2, ff.layers[0], ff.layers[2])
        a = self.attention(xh, ch, ch, residual=xh)
        a = ln(a, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self.feedforward(a, residual=xh)


class SelfAttentionBlock(nn.Module):
    def __init__(self, layer_num, model_dim=256, head_num=8, head_dim=64, norm_type="pre", act_fn="gelu", att_type="full", dropout=0.0):
        super().__init__()
        self.layers = nn.Sequential(*[
            GenericEncoderLayer(model_dim=model_dim, head_num=head_num, head_dim=head_dim, norm_type=norm_type, act_fn=act_fn,
                                att_type=att_type, att_mode="self", dropout=dropout) for _ in range(layer_num)])

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
