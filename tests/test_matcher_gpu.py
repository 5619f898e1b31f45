"""GPU parity of the matcher half: HIP kernels (through the C ABI) vs golden vectors / the oracle.
Index outputs must be bit-exact; scores / features within 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, compare_matches
from nerfmatch_amd import synth, ops, _lib
from nerfmatch_amd.matcher import NeRFMatcherMS, NeRFMatcherCoarse, PositionEncodingSine
from nerfmatch_amd.modules import PrecomputedBackbone
from nerfmatch_amd.modules.attention import GenericEncoderLayer
from oracle import matcher_oracle as mo

pytestmark = pytest.mark.gpu
TOL = 1e-4


def maxdiff(a, b):
    return (a.detach().cpu().float() - torch.as_tensor(b).float()).abs().max().item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("M,N,K,act,bias,res", [(77, 256, 256, 0, True, False), (4800, 256, 256, 2, True, True), (100, 128, 256, 1, True, False),
                                                 (33, 128, 128, 0, False, True), (1000, 256, 352, 0, True, False), (5, 96, 64, 2, False, False)])
def test_linear(gpu, built_lib, M, N, K, act, bias, res):
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K**-0.5)
    b = rnd(N, seed=3) if bias else None
    r = rnd(M, N, seed=4) if res else None
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref
    if res:
        ref = ref + r
    y = ops.linear(x.to(gpu), w.to(gpu), None if b is None else b.to(gpu), None if r is None else r.to(gpu), act)
    assert maxdiff(y, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,act,bias,res", [(301, 256, 256, 0, True, False), (4800, 768, 256, 0, False, False),
                                                 (129, 256, 352, 1, True, True), (77, 128, 264, 2, True, True),
                                                 (1000, 40, 24, 0, True, False), (19200, 256, 256, 2, True, True),
                                                 (203, 392, 256, 0, True, True), (33, 8, 8, 1, False, True), (40000, 512, 256, 0, True, False)])
def test_linear_bf16x3(gpu, built_lib, M, N, K, act, bias, res):
    """nm_linear_bf16x3 (split-bf16 MFMA, packed weights) against the fp32 torch reference: ragged M, N not a multiple of
    the 128-column tile, K % 16 == 8 tail, all epilogues."""
    x, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K**-0.5)
    b = rnd(N, seed=3) if bias else None
    r = rnd(M, N, seed=4) if res else None
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref
    if res:
        ref = ref + r
    ops.LINEAR_PRECISION = "bf16x3"
    try:
        y = ops.linear(x.to(gpu), w.to(gpu), None if b is None else b.to(gpu), None if r is None else r.to(gpu), act)
    finally:
        ops.LINEAR_PRECISION = "fp32"
    # measured against fp64 (scripts/lin_err.py): rms 4.4e-6, max 2.6e-5 over 1.5e7 outputs of magnitude ~1 (fp32 MFMA: 3.7e-6)
    assert maxdiff(y, ref) < 5e-5


@pytest.mark.parametrize("dim", [128, 256])
def test_layernorm(gpu, built_lib, dim):
    x, g, b = rnd(301, dim, seed=1, scale=3.0) + 0.5, 1 + 0.1 * rnd(dim, seed=2), 0.1 * rnd(dim, seed=3)
    y = ops.layernorm(x.to(gpu), g.to(gpu), b.to(gpu))
    assert maxdiff(y, F.layer_norm(x, (dim,), g, b)) < 1e-5


@pytest.mark.parametrize("B,L,S,H,D", [(1, 80, 96, 8, 32), (2, 200, 333, 8, 32), (1, 4800, 4800, 8, 32), (7, 25, 25, 8, 16), (3, 25, 25, 8, 32)])
def test_attention(gpu, built_lib, B, L, S, H, D):
    q, k, v = rnd(B, L, H * D, seed=1), rnd(B, S, H * D, seed=2), rnd(B, S, H * D, seed=3)
    scale = D**-0.5
    qg, kg, vg = q.to(gpu), k.to(gpu), v.to(gpu)
    out = ops.attention(qg, kg, vg, H, scale)
    if L * S > 4_000_000:  # reference on the GPU via torch ops in fp32 (too slow / big for the CPU oracle)
        ref = F.scaled_dot_product_attention(qg.view(B, L, H, D).transpose(1, 2), kg.view(B, S, H, D).transpose(1, 2),
                                             vg.view(B, S, H, D).transpose(1, 2), scale=scale).transpose(1, 2).reshape(B, L, H * D).cpu()
        sub = slice(0, L, 97)
        qs = q[:, sub].view(B, -1, H, D)
        sc = torch.einsum("blhd,bshd->blsh", qs * scale, k.view(B, S, H, D))
        ref_cpu = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D)).reshape(B, -1, H * D)
        assert maxdiff(out[:, sub], ref_cpu) < 2e-5
        assert maxdiff(out, ref) < 1e-4
    else:
        sc = torch.einsum("blhd,bshd->blsh", q.view(B, L, H, D) * scale, k.view(B, S, H, D))
        ref = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D)).reshape(B, L, H * D)
        assert maxdiff(out, ref) < 2e-5


@pytest.mark.parametrize("B,L,S", [(1, 4800, 4800), (2, 333, 200), (1, 100, 77)])
def test_attention_fp8_error_bound(gpu, built_lib, B, L, S, monkeypatch):
    """BASELINE config 5's throughput arithmetic (both contractions on ONE e4m3 MFMA per product block, csrc/attention_fp8.hip).
    NOT held to 1e-4.  e4m3 carries 3 mantissa bits (relative spacing 2^-3, rounding error <= 2^-4): the stated bound against the
    fp64 softmax attention is  max |error| <= 0.075 max|v|  (a row that one key dominates returns that key's value row as
    quantised: 2^-4 of its magnitude)  and  rms error <= 8 % of the output's rms  (measured with q, k, v ~ N(0,1): max 0.05 max|v|,
    rms 5-6 %); ragged sizes (S not a multiple of 64, L not a multiple of 128) and a dominating key included."""
    H, D = 8, 32
    q, k, v = rnd(B, L, H * D, seed=1), rnd(B, S, H * D, seed=2), rnd(B, S, H * D, seed=3)
    k[0, S // 2] = q[0, 5] * 3.0  # one key dominates query 5 in every head
    scale = D**-0.5
    monkeypatch.setattr(ops, "ATTENTION_PRECISION", "fp8")
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, scale).cpu()
    sub = slice(0, L, max(1, L // 150))
    sc = torch.einsum("blhd,bshd->blsh", q[:, sub].view(B, -1, H, D).double() * scale, k.view(B, S, H, D).double())
    ref = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D).double()).reshape(B, -1, H * D).float()
    err = (out[:, sub] - ref).abs()
    print(f"fp8 attention B={B} L={L} S={S}: max err {err.max():.3e}  rms err {err.pow(2).mean().sqrt():.3e}  (output rms {ref.pow(2).mean().sqrt():.3f})")
    assert torch.isfinite(out).all()
    assert err.max() < 0.075 * v.abs().max() and err.pow(2).mean().sqrt() < 0.08 * ref.pow(2).mean().sqrt()
    # and it is really the fp8 path: the split-bf16 kernel is ~1e-6 from the same reference
    monkeypatch.setattr(ops, "ATTENTION_PRECISION", "bf16x3")
    out2 = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, scale).cpu()
    assert (out2[:, sub] - ref).abs().max() < 1e-4 < err.max()


def test_attention_rescale_branch(gpu, built_lib):
    """Force the online-softmax running max to jump late in the key sequence (one key dominating every query)."""
    B, L, S, H, D = 1, 64, 256, 8, 32
    q, k, v = rnd(B, L, H * D, seed=1), rnd(B, S, H * D, seed=2), rnd(B, S, H * D, seed=3)
    k[0, 200] = q[0, 5] * 6.0  # key 200 (7th tile) matches query 5 strongly
    scale = D**-0.5
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, scale)
    sc = torch.einsum("blhd,bshd->blsh", q.view(B, L, H, D).double() * scale, k.view(B, S, H, D).double())
    ref = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D).double()).reshape(B, L, H * D)
    assert maxdiff(out, ref.float()) < 2e-5


def test_tokens_pe_fourier(gpu, built_lib):
    fx = load_golden("matcher_c2f")
    cfeat = fx["cfeat"]
    b, c, h, w = cfeat.shape
    pe = PositionEncodingSine(c).pe[0]
    assert maxdiff(pe[:, :h, :w], fx["pe_table"]) < 1e-6  # host sin/cos may differ by 1 ulp between CPUs
    tok = ops.nchw_to_tokens(cfeat.to(gpu), pe.to(gpu).contiguous())
    ref = (cfeat + fx["pe_table"][None]).flatten(-2).permute(0, 2, 1)
    assert maxdiff(tok, ref) < 1e-6
    assert maxdiff(ops.nchw_to_tokens(cfeat.to(gpu)), cfeat.flatten(-2).permute(0, 2, 1)) == 0
    pt3d = fx["pt3d"][0]
    feat = fx["pt_feat"][0]
    cat = ops.cat_fourier(feat.to(gpu), pt3d.to(gpu), 15)
    assert cat.shape[1] == 352
    assert maxdiff(cat[:, :256], feat) == 0
    assert maxdiff(cat[:, 256:349], fx["fourier_pt3d"][0]) < 1e-6
    assert float(cat[:, 349:].abs().max()) == 0.0
    # large arguments (metres x 2^14): accuracy of the fp64-reduced sin/cos
    big = torch.tensor([[123.456, -77.7, 301.25]])
    cat2 = ops.cat_fourier(torch.zeros(1, 256, device=gpu), big.to(gpu), 15)
    assert maxdiff(cat2[:, 256:349], mo.fourier_embed(big.double()).float()) < 2e-6


def load_layer(sd, prefix, layer):
    layer.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}, strict=True)
    return layer


def test_encoder_layers_vs_golden(gpu, built_lib):
    fx = load_golden("matcher_c2f")
    sd = synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"]))
    sa = load_layer(sd, "pt_sa.layers.0", GenericEncoderLayer(model_dim=256, head_dim=32, att_mode="self")).to(gpu)
    assert maxdiff(sa(fx["enc_self_in"].to(gpu)), fx["enc_self_out"]) < TOL
    ca = load_layer(sd, "coarse_former", GenericEncoderLayer(model_dim=256, context_dim=256, head_dim=32, att_mode="cross")).to(gpu)
    assert maxdiff(ca(fx["enc_self_in"].to(gpu), fx["pt_feat"].to(gpu)), fx["enc_cross_out"]) < TOL


@pytest.mark.parametrize("rows", [77, 128, 4800, 19333])
def test_encoder_tail_fused_vs_separate_launches(gpu, built_lib, rows, monkeypatch):
    """csrc/encoder_tail.hip (round 3): proj_out + residual + LayerNorm + feed-forward + residual of a pre-norm encoder layer as
    ONE launch against (a) the fp64 formula y = xh + W2 gelu(W1 LN2(xh + att Wo^T) + b1) + b2 (attention.py:229-241) and (b) the
    four separate launches it replaces; ragged row counts (tail workgroups), non-trivial LayerNorm parameters."""
    monkeypatch.setattr(ops, "LINEAR_PRECISION", "bf16x3")
    layer = GenericEncoderLayer(model_dim=256, head_dim=32, att_mode="self")
    sd = {}
    synth._encoder_layer(sd, np.random.default_rng(3), "L", 256)
    load_layer(sd, "L", layer).to(gpu)
    att, xh = rnd(rows, 256, seed=1), rnd(rows, 256, seed=2, scale=1.3)
    ff = layer.feedforward
    assert ops.encoder_tail_supported(256, 256, 256, ff.act)
    y = ops.encoder_tail(att.to(gpu), xh.to(gpu), layer.attention.proj_out[0].weight, layer.norm2, ff.layers[0], ff.layers[2]).cpu()
    # (b) the separate launches
    a = ops.linear(att.to(gpu), layer.attention.proj_out[0].weight, residual=xh.to(gpu))
    a = ops.layernorm(a, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    y_sep = ff(a, residual=xh.to(gpu)).cpu()
    # (a) fp64
    p = {k: v.double() for k, v in sd.items()}
    a64 = xh.double() + att.double() @ p["L.attention.proj_out.0.weight"].T
    a64 = F.layer_norm(a64, (256,), p["L.norm2.weight"], p["L.norm2.bias"], 1e-5)
    h64 = F.gelu(a64 @ p["L.feedforward.layers.0.weight"].T + p["L.feedforward.layers.0.bias"])
    y64 = xh.double() + h64 @ p["L.feedforward.layers.2.weight"].T + p["L.feedforward.layers.2.bias"]
    e_ref, e_sep = (y.double() - y64).abs().max().item(), (y - y_sep).abs().max().item()
    print(f"encoder tail rows={rows}: vs fp64 {e_ref:.2e}, vs separate launches {e_sep:.2e}")
    assert e_ref < 2e-5 and e_sep < 2e-5


def test_encoder_layers_vs_golden_bf16x3_fused_tail(gpu, built_lib):
    """The reference's encoder-layer fixtures through the fused tail (split-bf16 arithmetic): self and cross attention layers."""
    import nerfmatch_amd

    fx = load_golden("matcher_c2f")
    sd = synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"]))
    sa = load_layer(sd, "pt_sa.layers.0", GenericEncoderLayer(model_dim=256, head_dim=32, att_mode="self")).to(gpu)
    ca = load_layer(sd, "coarse_former", GenericEncoderLayer(model_dim=256, context_dim=256, head_dim=32, att_mode="cross")).to(gpu)
    nerfmatch_amd.set_precision("bf16x3")
    try:
        assert ops.encoder_tail_supported(256, 256, 256, sa.feedforward.act)
        assert maxdiff(sa(fx["enc_self_in"].to(gpu)), fx["enc_self_out"]) < TOL
        assert maxdiff(ca(fx["enc_self_in"].to(gpu), fx["pt_feat"].to(gpu)), fx["enc_cross_out"]) < TOL
    finally:
        nerfmatch_amd.set_precision("fp32")


def test_lsa_layer_vs_golden(gpu, built_lib):
    fx = load_golden("matcher_lsa")
    rng = np.random.default_rng(int(fx["weights_seed"]))
    sd = {}
    synth._encoder_layer(sd, rng, "L", 128)
    sd["L.attention.attend.scale"] = torch.as_tensor(fx["scale"])
    layer = load_layer(sd, "L", GenericEncoderLayer(model_dim=128, head_dim=16, att_type="lsa", att_mode="self")).to(gpu)
    assert maxdiff(layer(fx["x"].to(gpu)), fx["y"]) < TOL


def make_c2f(fx, gpu):
    m = NeRFMatcherMS(synth.matcher_config("c2f"))
    r = m.load_state_dict(synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"])), strict=False)
    assert not r.unexpected_keys and all(k.startswith("im_sa.") for k in r.missing_keys)
    m.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    return m.to(gpu).eval()


@pytest.mark.parametrize("tag,mutual,thr,masked", [("mut", True, 0.0, False), ("nomut", False, 0.0, False), ("mask", True, 0.0, True),
                                                     ("thr", True, None, False), ("empty", True, 0.5, False)])
def test_c2f_forward_vs_golden(gpu, built_lib, tag, mutual, thr, masked):
    fx = load_golden("matcher_c2f")
    m = make_c2f(fx, gpu)
    thr = fx["thr"] if thr is None else thr
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], fx["pt_feat"].shape[1]
    imm = fx["im_mask_partial"] if masked else torch.ones(1, M, dtype=torch.bool)
    ptm = fx["pt_mask_partial"] if masked else torch.ones(1, N, dtype=torch.bool)
    data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=imm.to(gpu), pt3d=fx["pt3d"].to(gpu), pt_feat=fx["pt_feat"].to(gpu),
                pt_mask=ptm.to(gpu), pt2d=fx["pt2d"].to(gpu))
    assert m.forward(data, ret_feats=True, mutual=mutual, match_thres=thr) is None  # mutates in place like the reference
    b, i, j = data["match_ids"]
    assert b.dtype == torch.int64
    assert torch.equal(b.cpu(), fx[f"{tag}_b_ids"]) and torch.equal(i.cpu(), fx[f"{tag}_i_ids"]) and torch.equal(j.cpu(), fx[f"{tag}_j_ids"])
    assert data["mconf"].shape == fx[f"{tag}_mconf"].shape
    if len(b):
        assert maxdiff(data["mconf"], fx[f"{tag}_mconf"]) < TOL
        assert maxdiff(data["expec_f"], fx[f"{tag}_expec_f"]) < TOL
        assert maxdiff(data["mpt2d_f"], fx[f"{tag}_mpt2d_f"]) < 10 * TOL  # pixels (expec * 5)
        assert maxdiff(data["mpt2d_c"], fx[f"{tag}_mpt2d_c"]) == 0
        assert maxdiff(data["mpt3d"], fx[f"{tag}_mpt3d"]) == 0
    else:
        assert data["expec_f"].shape == (0, 3) and data["mpt2d_f"].shape == (0, 2)
    if tag in ("mut", "mask"):
        assert maxdiff(data["conf_matrix"], fx[f"{tag}_conf"]) < TOL
        assert maxdiff(data["im_cfeat"], fx[f"{tag}_im_cfeat"]) < TOL
        assert maxdiff(data["pt_cfeat"], fx[f"{tag}_pt_cfeat"]) < TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("tag,mutual,thr,masked", [("mut", True, 0.0, False), ("nomut", False, 0.0, False), ("mask", True, 0.0, True),
                                                     ("thr", True, None, False)])
def test_c2f_forward_peaked_vs_reference(gpu, built_lib, tag, mutual, thr, masked, precision):
    """Round 3, VERDICT r2 item 1 (ii): the PEAKED-confidence regime (tests/golden/matcher_peaked.npz, generated by the reference
    itself): 320 image tokens x 352 points (11 key tiles, 3 GEMM row tiles), ~90 % of the rows mutual matches with a row maximum
    near 1, the rest diffuse.  Indices bit-exact, scores within 1e-4, BOTH arithmetic paths of the contractions."""
    import nerfmatch_amd

    fx = load_golden("matcher_peaked")
    m = NeRFMatcherMS(synth.matcher_config("c2f"))
    r = m.load_state_dict(synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"]), temperature=float(fx["temperature"]), style="aligned"), strict=False)
    assert not r.unexpected_keys and all(k.startswith("im_sa.") for k in r.missing_keys)
    m.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    m.to(gpu).eval()
    thr = float(fx["thr"]) if thr is None else thr
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], fx["pt_feat"].shape[1]
    imm = fx["im_mask_partial"] if masked else torch.ones(1, M, dtype=torch.bool)
    ptm = fx["pt_mask_partial"] if masked else torch.ones(1, N, dtype=torch.bool)
    data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=imm.to(gpu), pt3d=fx["pt3d"].to(gpu), pt_feat=fx["pt_feat"].to(gpu),
                pt_mask=ptm.to(gpu), pt2d=fx["pt2d"].to(gpu))
    nerfmatch_amd.set_precision(precision)
    try:
        m.forward(data, ret_feats=True, mutual=mutual, match_thres=thr)
    finally:
        nerfmatch_amd.set_precision("fp32")
    b, i, j = data["match_ids"]
    e_conf = maxdiff(data["mconf"], fx[f"{tag}_mconf"]) if len(b) == len(fx[f"{tag}_b_ids"]) else float("nan")
    print(f"peaked {tag} [{precision}]: {len(b)} matches (reference {len(fx[f'{tag}_b_ids'])}), mconf err {e_conf:.2e}")
    assert torch.equal(b.cpu(), fx[f"{tag}_b_ids"]) and torch.equal(i.cpu(), fx[f"{tag}_i_ids"]) and torch.equal(j.cpu(), fx[f"{tag}_j_ids"])
    assert e_conf < TOL
    assert maxdiff(data["expec_f"], fx[f"{tag}_expec_f"]) < TOL
    assert maxdiff(data["mpt2d_f"], fx[f"{tag}_mpt2d_f"]) < 10 * TOL
    assert maxdiff(data["mpt3d"], fx[f"{tag}_mpt3d"]) == 0
    if tag in ("mut", "mask"):
        assert maxdiff(data["conf_matrix"], fx[f"{tag}_conf"]) < TOL
    if tag == "mut":
        assert maxdiff(data["im_cfeat"], fx["mut_im_cfeat"]) < TOL and maxdiff(data["pt_cfeat"], fx["mut_pt_cfeat"]) < TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("tag,mutual", [("mut", True), ("nomut", False)])
def test_coarse_forward_peaked_vs_reference(gpu, built_lib, tag, mutual, precision, monkeypatch):
    """The coarse-only model on the peaked fixture (row maxima ~0.999), against the reference's run."""
    monkeypatch.setattr(ops, "MATCH_PRECISION", precision)
    fx, fxc = load_golden("matcher_peaked"), load_golden("matcher_peaked_coarse")
    m = NeRFMatcherCoarse(synth.matcher_config("coarse"))
    m.load_state_dict(synth.matcher_state_dict("coarse", temperature=float(fxc["temperature"])), strict=False)
    m.backbone = PrecomputedBackbone(fx["cfeat"].to(gpu), 256)
    m.to(gpu).eval()
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], fx["pt_feat"].shape[1]
    data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu),
                pt3d=torch.zeros(1, N, 3, device=gpu), pt_feat=fx["pt_feat"].to(gpu), pt_mask=torch.ones(1, N, dtype=torch.bool, device=gpu), pt2d=None)
    m.forward(data, mutual=mutual)
    b, i, j = data["match_ids"]
    assert torch.equal(i.cpu(), fxc[f"{tag}_i_ids"]) and torch.equal(j.cpu(), fxc[f"{tag}_j_ids"]) and torch.equal(b.cpu(), fxc[f"{tag}_b_ids"])
    assert maxdiff(data["mconf"], fxc[f"{tag}_mconf"]) < TOL
    if mutual:
        assert maxdiff(data["conf_matrix"], fxc["conf"]) < TOL


@pytest.mark.parametrize("tag,mutual", [("mut", True), ("nomut", False)])
def test_coarse_forward_vs_golden(gpu, built_lib, tag, mutual):
    fx = load_golden("matcher_coarse")
    m = NeRFMatcherCoarse(synth.matcher_config("coarse"))
    m.load_state_dict(synth.matcher_state_dict("coarse"), strict=False)
    m.backbone = PrecomputedBackbone(fx["cfeat"].to(gpu), 256)
    m.to(gpu).eval()
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], fx["pt_feat"].shape[1]
    data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu),
                pt3d=torch.zeros(1, N, 3, device=gpu), pt_feat=fx["pt_feat"].to(gpu), pt_mask=torch.ones(1, N, dtype=torch.bool, device=gpu), pt2d=None)
    out = m.forward(data, mutual=mutual)
    assert out is data  # the coarse model returns the dict
    b, i, j = data["match_ids"]
    assert torch.equal(i.cpu(), fx[f"{tag}_i_ids"]) and torch.equal(j.cpu(), fx[f"{tag}_j_ids"]) and torch.equal(b.cpu(), fx[f"{tag}_b_ids"])
    assert maxdiff(data["mconf"], fx[f"{tag}_mconf"]) < TOL
    if mutual:
        assert maxdiff(data["conf_matrix"], fx["conf"]) < TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("M,N", [(4800, 4800), (3600, 3600), (1000, 777), (1000, 776)])
def test_dual_softmax_full_size_vs_oracle(gpu, built_lib, M, N, precision, monkeypatch):
    """BASELINE config C2 shapes: indices bit-exact on planted (well separated) correspondences, scores within 1e-4;
    on the unplanted remainder (adversarial near-ties) every differing row must be an oracle tie (<= 2e-5 relative).  Both arithmetic paths of
    the similarity GEMM (N = 777 is not a multiple of 8: the bf16x3 request falls back to the fp32 GEMM)."""
    monkeypatch.setattr(ops, "MATCH_PRECISION", precision)
    im, pt = synth.separated_features(M, N, 256, seed=2)
    g = torch.Generator().manual_seed(9)
    n_plant = min(M, N) // 2
    perm = torch.randperm(N, generator=g)[:n_plant]
    pt[perm] = im[:n_plant] + 0.02 * torch.randn(n_plant, 256, generator=g)
    for mutual in (True, False):
        r = ops.dual_softmax_match(im.to(gpu), pt.to(gpu), 10.0, threshold=0.0, mutual=mutual, want_conf=True)
        conf, _, _ = mo.coarse_matching(im[None], pt[None], torch.tensor(10.0))
        ids, mconf = mo.mutual_matches(conf, mutual=mutual, threshold=0.0)
        assert maxdiff(r["conf"], conf[0]) < TOL
        gi, gj = r["i_ids"].cpu(), r["j_ids"].cpu()
        ref = dict(zip(ids[1].tolist(), ids[2].tolist()))
        got = dict(zip(gi.tolist(), gj.tolist()))
        planted_ok = all(got.get(i) == int(perm[i]) and ref.get(i) == int(perm[i]) for i in range(n_plant))
        assert planted_ok
        # identical index lists; a differing row must be a numerical tie in the ORACLE's own conf values (conftest.tie_excused)
        compare_matches((ids[1], ids[2]), (gi, gj), conf[0], mutual, f"dual-softmax {M}x{N} {precision} mutual={mutual}")
        assert torch.all(gi[1:] > gi[:-1])  # sorted by image token, one match per token


def _batch_match(im, pt, scale, gpu, fused, monkeypatch, **kw):
    monkeypatch.setattr(ops, "MATCH_PRECISION", "bf16x3")
    monkeypatch.setattr(ops, "MATCH_FUSED", fused)
    r = ops.dual_softmax_match_batch(im.to(gpu), pt.to(gpu), scale, want_conf=False, **{k: (v.to(gpu) if torch.is_tensor(v) else v) for k, v in kw.items()})
    cnt = r["count"].cpu().tolist()
    return [(r["i_ids"][b, :c].cpu(), r["j_ids"][b, :c].cpu(), r["mconf"][b, :c].cpu()) for b, c in enumerate(cnt)]


@pytest.mark.parametrize("M,N", [(4800, 4800), (3600, 3600), (1000, 776), (300, 333), (130, 70)])
@pytest.mark.parametrize("mutual", [True, False])
def test_fused_match_without_sim_vs_oracle(gpu, built_lib, M, N, mutual, monkeypatch):
    """csrc/match_fused.hip (round 3): a BATCH of pairs matched without the similarity matrix in HBM -- against the oracle (ids
    identical; a differing row must be an oracle tie, conftest.tie_excused; scores 1e-4) and against the per-pair kernels that
    materialise sim.  Ragged tiles (M, N not multiples of 128, N odd), two different pairs in the batch, planted
    correspondences, threshold."""
    g = torch.Generator().manual_seed(9)
    ims, pts, perms = [], [], []
    for b in range(2):
        im, pt = synth.separated_features(M, N, 256, seed=2 + b)
        n_plant = min(M, N) // 2
        perm = torch.randperm(N, generator=g)[:n_plant]
        pt[perm] = im[:n_plant] + 0.02 * torch.randn(n_plant, 256, generator=g)
        ims.append(im); pts.append(pt); perms.append(perm)
    im, pt = torch.stack(ims), torch.stack(pts)
    for thr in (0.0, 0.3):
        got = _batch_match(im, pt, 10.0, gpu, True, monkeypatch, threshold=thr, mutual=mutual)
        old = _batch_match(im, pt, 10.0, gpu, False, monkeypatch, threshold=thr, mutual=mutual)
        for b in range(2):
            conf, _, _ = mo.coarse_matching(im[b:b + 1], pt[b:b + 1], torch.tensor(10.0))
            ids, mconf = mo.mutual_matches(conf, mutual=mutual, threshold=thr)
            gi, gj, gc = got[b]
            n_plant = len(perms[b])
            if thr == 0.0:
                d = dict(zip(gi.tolist(), gj.tolist()))
                assert all(d.get(i) == int(perms[b][i]) for i in range(n_plant))
            compare_matches((ids[1], ids[2]), (gi, gj), conf[0], mutual, f"fused {M}x{N} pair {b} mutual={mutual} thr={thr}")
            assert torch.all(gi[1:] > gi[:-1])
            ref = dict(zip(ids[1].tolist(), mconf.tolist()))
            both = [k for k, i in enumerate(gi.tolist()) if i in ref]
            if both:
                assert maxdiff(gc[both], torch.tensor([ref[int(gi[k])] for k in both])) < TOL
            # the per-pair path agrees too (same GEMM arithmetic, soft-max shifted differently: rounding only)
            oi, oj, oc = old[b]
            compare_matches((oi, oj), (gi, gj), conf[0], mutual, f"fused vs sim-in-HBM {M}x{N} pair {b}")


@pytest.mark.parametrize("C", [64, 128, 512])
def test_fused_match_other_channel_counts(gpu, built_lib, C, monkeypatch):
    """The fused path's other instantiations (C = 64, 128, 512: 4, 8, 32 K-steps; C = 256 is covered above): ids against the oracle."""
    M, N = 300, 270
    im, pt = synth.separated_features(M, N, C, seed=11)
    pt[:100] = im[:100] + 0.02 * torch.randn(100, C, generator=torch.Generator().manual_seed(2))
    for mutual in (True, False):
        got = _batch_match(im[None], pt[None], 10.0, gpu, True, monkeypatch, mutual=mutual, threshold=0.1)[0]
        conf, _, _ = mo.coarse_matching(im[None], pt[None], torch.tensor(10.0))
        ids, mconf = mo.mutual_matches(conf, mutual=mutual, threshold=0.1)
        compare_matches((ids[1], ids[2]), (got[0], got[1]), conf[0], mutual, f"fused C={C} mutual={mutual}")
        ref = dict(zip(ids[1].tolist(), mconf.tolist()))
        both = [k for k, i in enumerate(got[0].tolist()) if i in ref]
        assert both and maxdiff(got[2][both], torch.tensor([ref[int(got[0][k])] for k in both])) < TOL


def test_fused_match_masks_ties_and_fallback(gpu, built_lib, monkeypatch):
    """Masks (partially and entirely masked sides: uniform soft-max, exact ties -> the tie pass picks the reference's first
    column), duplicated points (exact ties between columns) and the documented fall-back for |scale| log2 e > 60."""
    M, N = 200, 170
    im, pt = synth.separated_features(M, N, 256, seed=4)
    pt[:60] = im[:60] + 0.01 * torch.randn(60, 256, generator=torch.Generator().manual_seed(1))
    pt[100] = pt[7]   # two identical points: columns 7 and 100 tie exactly in every row
    pt[101] = pt[7]
    imm = torch.ones(M, dtype=torch.bool); imm[20:31] = False
    ptm = torch.ones(N, dtype=torch.bool); ptm[3:9] = False
    cases = [dict(), dict(im_mask=imm[None], pt_mask=ptm[None]), dict(pt_mask=torch.zeros(1, N, dtype=torch.bool)),
             dict(im_mask=torch.zeros(1, M, dtype=torch.bool))]
    for kw in cases:
        for mutual in (True, False):
            got = _batch_match(im[None], pt[None], 10.0, gpu, True, monkeypatch, mutual=mutual, **kw)[0]
            conf, _, _ = mo.coarse_matching(im[None], pt[None], torch.tensor(10.0), kw.get("im_mask"), kw.get("pt_mask"))
            ids, mconf = mo.mutual_matches(conf, mutual=mutual)
            assert torch.equal(got[0], ids[1]) and torch.equal(got[1], ids[2]), (list(kw), mutual, len(got[0]), len(ids[1]))
            if len(mconf):
                assert maxdiff(got[2], mconf) < TOL
    # a temperature the fixed shift cannot take: the batch call silently uses the per-pair kernels, same answer
    got = _batch_match(im[None], pt[None], 50.0, gpu, True, monkeypatch, mutual=True)[0]
    conf, _, _ = mo.coarse_matching(im[None], pt[None], torch.tensor(50.0))
    ids, _ = mo.mutual_matches(conf, mutual=True)
    assert torch.equal(got[0], ids[1]) and torch.equal(got[1], ids[2])


def test_c2f_batch_of_two_equals_singles(gpu, built_lib):
    """B = 2 (two queries per step) gives the per-query results of two B = 1 calls, with b_ids labelling the rows."""
    fx = load_golden("matcher_c2f")
    m = make_c2f(fx, gpu)
    cf2 = torch.cat([fx["cfeat"], fx["cfeat"].flip(-1)]).to(gpu)
    ff2 = torch.cat([fx["ffeat"], fx["ffeat"].flip(-1)]).to(gpu)
    pf2 = torch.cat([fx["pt_feat"], fx["pt_feat"].roll(3, 1)]).to(gpu)
    p32 = torch.cat([fx["pt3d"], fx["pt3d"].roll(3, 1)]).to(gpu)
    M, N = cf2.shape[2] * cf2.shape[3], pf2.shape[1]
    outs = []
    for b in range(2):
        m.backbone = PrecomputedBackbone((cf2[b:b + 1].contiguous(), ff2[b:b + 1].contiguous()), [256, 128])
        d = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu), pt3d=p32[b:b + 1].contiguous(),
                 pt_feat=pf2[b:b + 1].contiguous(), pt_mask=torch.ones(1, N, dtype=torch.bool, device=gpu), pt2d=fx["pt2d"].to(gpu))
        m.forward(d, mutual=True)
        outs.append(d)
    m.backbone = PrecomputedBackbone((cf2, ff2), [256, 128])
    d = dict(image=torch.zeros(2, 3, 8, 8, device=gpu), im_mask=torch.ones(2, M, dtype=torch.bool, device=gpu), pt3d=p32, pt_feat=pf2,
             pt_mask=torch.ones(2, N, dtype=torch.bool, device=gpu), pt2d=fx["pt2d"].expand(2, -1, -1).contiguous().to(gpu))
    m.forward(d, mutual=True)
    bb, ii, jj = d["match_ids"]
    for b in range(2):
        sel = bb == b
        assert torch.equal(ii[sel], outs[b]["match_ids"][1]) and torch.equal(jj[sel], outs[b]["match_ids"][2])
        assert maxdiff(d["mconf"][sel], outs[b]["mconf"].cpu()) < 1e-6
        assert maxdiff(d["expec_f"][sel], outs[b]["expec_f"].cpu()) < 1e-5
        assert maxdiff(d["mpt2d_f"][d["m_bids"] == b], outs[b]["mpt2d_f"].cpu()) < 1e-4


def test_temp_type_div_and_masks_all_false(gpu, built_lib):
    """temp_type 'div' (LoFTR temperature 0.1, c2f_trainer.py:101-106) and a fully masked point set."""
    im, pt = synth.separated_features(96, 80, 256, seed=4)
    pt[:40] = im[:40] + 0.01 * torch.randn(40, 256, generator=torch.Generator().manual_seed(1))
    conf, _, _ = mo.coarse_matching(im[None], pt[None], torch.tensor(0.1), temp_type="div")
    ids, mconf = mo.mutual_matches(conf, mutual=True)
    r = ops.dual_softmax_match(im.to(gpu), pt.to(gpu), 1.0 / 0.1, mutual=True)
    assert torch.equal(r["i_ids"].cpu(), ids[1]) and torch.equal(r["j_ids"].cpu(), ids[2])
    assert maxdiff(r["conf"], conf[0]) < TOL
    # everything masked: sim = -1e9 everywhere -> uniform softmaxes, conf = 1/(M*N) for all; mutual ties: the
    # reference picks the first column of every row
    m0 = torch.zeros(80, dtype=torch.bool)
    conf0, _, _ = mo.coarse_matching(im[None], pt[None], torch.tensor(10.0), None, m0[None])
    ids0, _ = mo.mutual_matches(conf0, mutual=True)
    r0 = ops.dual_softmax_match(im.to(gpu), pt.to(gpu), 10.0, pt_mask=m0.to(gpu), mutual=True)
    assert torch.equal(r0["i_ids"].cpu(), ids0[1]) and torch.equal(r0["j_ids"].cpu(), ids0[2])
    assert maxdiff(r0["conf"], conf0[0]) < 1e-8


def test_multi_pair_matches_per_pair_loop(gpu, built_lib):
    """pt3d (B,k,N,3): forward_multi_pair concatenates the per-reference-frame matches (c2f_trainer.py:371-427)."""
    fx = load_golden("matcher_c2f")
    m = make_c2f(fx, gpu)
    N = fx["pt_feat"].shape[1]
    M = fx["cfeat"].shape[2] * fx["cfeat"].shape[3]
    pf = torch.stack([fx["pt_feat"][0], fx["pt_feat"][0].roll(5, 0), fx["pt_feat"][0].flip(0)])[None].to(gpu)   # (1,3,N,256)
    p3 = torch.stack([fx["pt3d"][0], fx["pt3d"][0].roll(5, 0), fx["pt3d"][0].flip(0)])[None].to(gpu)
    common = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu), pt2d=fx["pt2d"].to(gpu))
    data = dict(common, pt3d=p3, pt_feat=pf, pt_mask=torch.ones(1, 3, N, dtype=torch.bool, device=gpu))
    m.forward(data, mutual=True)
    parts = []
    for k in range(3):
        d = dict(common, pt3d=p3[:, k].contiguous(), pt_feat=pf[:, k].contiguous(), pt_mask=torch.ones(1, N, dtype=torch.bool, device=gpu))
        m.forward(d, mutual=True)
        parts.append(d)
    assert data["mpt3d"].shape[0] == sum(p["mpt3d"].shape[0] for p in parts) > 0
    assert maxdiff(data["mpt3d"], torch.cat([p["mpt3d"] for p in parts]).cpu()) == 0
    assert maxdiff(data["mpt2d_f"], torch.cat([p["mpt2d_f"] for p in parts]).cpu()) < 1e-5
    assert maxdiff(data["mconf"], torch.cat([p["mconf"] for p in parts]).cpu()) < 1e-7
    # and the first pair is the golden single-pair case
    assert torch.equal(parts[0]["match_ids"][2].cpu(), fx["mut_j_ids"])


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_multi_pair_vs_reference(gpu, built_lib, precision):
    """forward_multi_pair against the reference's own run (tests/golden/matcher_multipair.npz: B = 2 queries x k = 3 reference
    frames, one point set partially masked; nerfmatch_c2f_trainer.py:371-427, nerfmatch_coarse_trainer.py:290-336): same
    matches in the same (frame-major) concatenation order, although the image side is evaluated once and the k point sets go
    through the kernels as one batch here."""
    import nerfmatch_amd

    fx = load_golden("matcher_multipair")
    B, k, N = fx["pt3d"].shape[:3]
    M = fx["cfeat"].shape[2] * fx["cfeat"].shape[3]
    m = NeRFMatcherMS(synth.matcher_config("c2f"))
    m.load_state_dict(synth.matcher_state_dict("c2f", seed=int(fx["weights_seed"])), strict=False)
    m.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    m.to(gpu).eval()
    mini = NeRFMatcherCoarse(synth.matcher_config("coarse"))
    mini.load_state_dict(synth.matcher_state_dict("coarse", seed=int(fx["weights_seed"])), strict=False)
    mini.backbone = PrecomputedBackbone(fx["cfeat"].to(gpu), 256)
    mini.to(gpu).eval()
    mk = lambda: dict(image=torch.zeros(B, 3, 8, 8, device=gpu), im_mask=fx["im_mask"].to(gpu), pt3d=fx["pt3d"].to(gpu), pt_feat=fx["pt_feat"].to(gpu),
                      pt_mask=fx["pt_mask"].to(gpu), pt2d=fx["pt2d"].to(gpu))
    nerfmatch_amd.set_precision(precision)
    try:
        for tag, mutual in (("mut", True), ("nomut", False)):
            data = mk()
            assert m.forward(data, mutual=mutual) is None
            assert torch.equal(data["m_bids"].cpu(), fx[f"c2f_{tag}_m_bids"])
            assert maxdiff(data["mpt3d"], fx[f"c2f_{tag}_mpt3d"]) == 0 and maxdiff(data["mpt2d_c"], fx[f"c2f_{tag}_mpt2d_c"]) == 0
            assert maxdiff(data["mconf"], fx[f"c2f_{tag}_mconf"]) < TOL
            assert maxdiff(data["mpt2d_f"], fx[f"c2f_{tag}_mpt2d_f"]) < 5 * TOL  # pixels: expec_f (1e-4) x 5
            data = mk()
            assert mini.forward(data, mutual=mutual) is data
            b, i, j = data["match_ids"]
            assert torch.equal(b.cpu(), fx[f"coarse_{tag}_b_ids"]) and torch.equal(i.cpu(), fx[f"coarse_{tag}_i_ids"]) and torch.equal(j.cpu(), fx[f"coarse_{tag}_j_ids"])
            assert maxdiff(data["mconf"], fx[f"coarse_{tag}_mconf"]) < TOL
    finally:
        nerfmatch_amd.set_precision("fp32")


# ----------------------------------------------------------------------------- bf16x3 attention
@pytest.fixture
def attn_bf16x3():
    """Both matcher contractions on the split-bf16 path: attention (nm_attention_ex) and nn.Linear (nm_linear_bf16x3)."""
    ops.ATTENTION_PRECISION = "bf16x3"
    ops.LINEAR_PRECISION = "bf16x3"
    ops.MATCH_PRECISION = "bf16x3"
    yield
    ops.ATTENTION_PRECISION = "fp32"
    ops.LINEAR_PRECISION = "fp32"
    ops.MATCH_PRECISION = "fp32"


@pytest.mark.parametrize("B,L,S", [(1, 80, 96), (2, 200, 333), (1, 4800, 4800), (1, 100, 32), (1, 70, 33), (2, 130, 100), (3, 65, 160)])
def test_attention_bf16x3(gpu, built_lib, attn_bf16x3, B, L, S):
    H, D = 8, 32
    q, k, v = rnd(B, L, H * D, seed=1), rnd(B, S, H * D, seed=2), rnd(B, S, H * D, seed=3)
    scale = D**-0.5
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, scale)
    sub = slice(0, L, max(1, L // 50))
    sc = torch.einsum("blhd,bshd->blsh", q[:, sub].view(B, -1, H, D).double() * scale, k.view(B, S, H, D).double())
    ref = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D).double()).reshape(B, -1, H * D).float()
    err = maxdiff(out[:, sub], ref)
    print(f"bf16x3 attention {B}x{L}x{S}: max err {err:.2e}")
    assert err < 3e-5


def test_attention_bf16x3_rescale_branch(gpu, built_lib, attn_bf16x3):
    B, L, S, H, D = 1, 64, 256, 8, 32
    q, k, v = rnd(B, L, H * D, seed=1), rnd(B, S, H * D, seed=2), rnd(B, S, H * D, seed=3)
    k[0, 200] = q[0, 5] * 6.0
    scale = D**-0.5
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, scale)
    sc = torch.einsum("blhd,bshd->blsh", q.view(B, L, H, D).double() * scale, k.view(B, S, H, D).double())
    ref = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D).double()).reshape(B, L, H * D)
    assert maxdiff(out, ref.float()) < 1e-4  # |score| ~ 200 here: the split's relative error shows in the exponent


@pytest.mark.parametrize("where", [40, 200, 230, 255])
def test_attention_bf16x3_late_maximum(gpu, built_lib, attn_bf16x3, where):
    """The third-generation kernel raises its running maximum only when a tile's probabilities near the fp32 range: a score far
    above everything seen so far, in an even or odd tile of the two-tile loop, in the first or the last tile, and rows whose
    scores are all very negative next to rows that are not."""
    B, L, S, H, D = 1, 96, 256, 8, 32
    q, k, v = rnd(B, L, H * D, seed=1), rnd(B, S, H * D, seed=2), rnd(B, S, H * D, seed=3)
    k[0, where] = q[0, 7] * 9.0          # score ~ +300 for query 7 at key `where`, large and mixed for the others
    q[0, 11] = -q[0, 11].abs() * 20.0     # one query whose scores are huge in magnitude and mostly of one sign
    scale = D**-0.5
    out = ops.attention(q.to(gpu), k.to(gpu), v.to(gpu), H, scale)
    sc = torch.einsum("blhd,bshd->blsh", q.view(B, L, H, D).double() * scale, k.view(B, S, H, D).double())
    ref = torch.einsum("blsh,bshd->blhd", torch.softmax(sc, 2), v.view(B, S, H, D).double()).reshape(B, L, H * D)
    assert torch.isfinite(out).all()
    assert maxdiff(out, ref.float()) < 2e-4  # scores of several hundred: the operand split's relative error sits in the exponent


@pytest.mark.parametrize("tag,mutual,masked", [("mut", True, False), ("nomut", False, False), ("mask", True, True)])
def test_c2f_forward_bf16x3_attention_vs_golden(gpu, built_lib, attn_bf16x3, tag, mutual, masked):
    """The whole c2f forward with split-bf16 attention keeps the golden indices and the 1e-4 tolerance."""
    fx = load_golden("matcher_c2f")
    m = make_c2f(fx, gpu)
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], fx["pt_feat"].shape[1]
    imm = fx["im_mask_partial"] if masked else torch.ones(1, M, dtype=torch.bool)
    ptm = fx["pt_mask_partial"] if masked else torch.ones(1, N, dtype=torch.bool)
    data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=imm.to(gpu), pt3d=fx["pt3d"].to(gpu), pt_feat=fx["pt_feat"].to(gpu),
                pt_mask=ptm.to(gpu), pt2d=fx["pt2d"].to(gpu))
    m.forward(data, ret_feats=True, mutual=mutual, match_thres=0.0)
    b, i, j = data["match_ids"]
    assert torch.equal(i.cpu(), fx[f"{tag}_i_ids"]) and torch.equal(j.cpu(), fx[f"{tag}_j_ids"])
    assert maxdiff(data["mconf"], fx[f"{tag}_mconf"]) < TOL and maxdiff(data["expec_f"], fx[f"{tag}_expec_f"]) < TOL
    if tag in ("mut", "mask"):
        assert maxdiff(data["conf_matrix"], fx[f"{tag}_conf"]) < TOL and maxdiff(data["im_cfeat"], fx[f"{tag}_im_cfeat"]) < TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_shared_self_attention_batched_equals_separate(gpu, built_lib, precision, monkeypatch):
    """`im_sa_type: share` with as many image tokens as points: both sets go through the self-attention block as one batch of
    2B sequences.  Every kernel of the block works per row / per sequence, so the results are those of the two separate
    passes, bit for bit."""
    fx = load_golden("matcher_c2f")
    m = make_c2f(fx, gpu)
    if precision == "bf16x3":
        monkeypatch.setattr(ops, "LINEAR_PRECISION", "bf16x3")
        monkeypatch.setattr(ops, "ATTENTION_PRECISION", "bf16x3")
        monkeypatch.setattr(ops, "MATCH_PRECISION", "bf16x3")
    g = torch.Generator().manual_seed(21)
    B, h, w = 2, 8, 16
    N = h * w
    cf = torch.randn(B, 256, h, w, generator=g).to(gpu)
    ff = torch.randn(B, 128, 4 * h, 4 * w, generator=g).to(gpu)
    m.backbone = PrecomputedBackbone((cf, ff), [256, 128])
    pt_feat = torch.relu(torch.randn(B, N, 256, generator=g))
    pt_feat[:, :40] = torch.relu(cf.flatten(-2).permute(0, 2, 1)[:, :40].cpu()) + 0.05 * torch.randn(B, 40, 256, generator=g)
    pt_feat, pt3d = pt_feat.to(gpu), (torch.randn(B, N, 3, generator=g) * 2).to(gpu)
    img = torch.zeros(B, 3, 8 * h, 8 * w, device=gpu)
    assert m._shared_sa_batchable(cf, pt_feat)
    got = m.forward_match(img, pt_feat, pt3d, mutual=True, ret_feats=True)
    monkeypatch.setattr(type(m), "_shared_sa_batchable", lambda self, c, p: False)
    want = m.forward_match(img, pt_feat, pt3d, mutual=True, ret_feats=True)
    assert got["pred_num"] > 20
    for a, b in zip(got["match_ids"], want["match_ids"]):
        assert torch.equal(a, b)
    for k in ("conf_matrix", "mconf", "expec_f", "im_cfeat", "pt_cfeat"):
        assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize("mode,B,L,S", [("self", 2, 96, 96), ("self", 1, 4800, 4800), ("cross", 2, 80, 96), ("cross", 1, 200, 4800),
                                        ("self", 3, 160, 160), ("cross", 2, 65, 128), ("cross", 5, 4800, 96)])
def test_projection_fused_into_attention_operands(gpu, built_lib, monkeypatch, mode, B, L, S):
    """nm_linear_qkv_bf16x3 + nm_attention_presplit (keys / values written by the projection GEMM straight into the attention
    kernel's pre-split operand slots; the value chunks through the transposed MFMA product) against the unfused sequence
    projection GEMM -> kv_presplit_kernel -> attention: same arithmetic, so the same numbers."""
    from nerfmatch_amd.modules.attention import MultiHeadAttention

    monkeypatch.setattr(ops, "LINEAR_PRECISION", "bf16x3")
    monkeypatch.setattr(ops, "ATTENTION_PRECISION", "bf16x3")
    torch.manual_seed(5)
    mha = MultiHeadAttention(256, head_num=8, head_dim=32).to(gpu).eval()
    x = torch.randn(B, L, 256, device=gpu)
    ctx = x if mode == "self" else torch.randn(B, S, 256, device=gpu)
    assert ops.projected_attention_supported(256, 8, 32, L, S)
    got = mha(x, ctx, ctx, residual=x)
    monkeypatch.setattr(ops, "projected_attention_supported", lambda *a: False)
    want = mha(x, ctx, ctx, residual=x)
    assert maxdiff(got, want.cpu()) <= 1e-6 * float(want.abs().max()), maxdiff(got, want.cpu())


def test_small_grid_gemm_is_bit_identical_to_the_ring_kernel(gpu, built_lib):
    """Round 5: launches whose workgroups are all resident at once (one query: 4800 rows = 38 row tiles) take the small-grid form of the
    split-bf16 GEMM (every request issued up front, csrc/gemm_bf16.hip); it runs the same products in the same order as the ring kernel,
    so a row's result must not depend on how many other rows the launch holds -- what a query returns is the same bits at batch 1 and 16."""
    import nerfmatch_amd
    from nerfmatch_amd import _lib
    from nerfmatch_amd.modules.attention import GenericEncoderLayer

    g = torch.Generator().manual_seed(11)
    nerfmatch_amd.set_precision("bf16x3")
    try:
        for K, N in ((256, 256), (256, 128), (128, 128), (128, 384), (256, 768)):
            w = (torch.randn(N, K, generator=g) / K**0.5).to(gpu)
            b = torch.randn(N, generator=g).to(gpu)
            big = torch.randn(160000, K, generator=g).to(gpu)
            res = torch.randn(160000, N, generator=g).to(gpu)
            for m in (4800, 150, 3750):
                for kw in (dict(), dict(bias=b), dict(bias=b, act=_lib.NM_ACT_GELU), dict(bias=b, residual=True)):
                    kw_s, kw_b = dict(kw), dict(kw)
                    if kw.get("residual"):
                        kw_s["residual"], kw_b["residual"] = res[:m].contiguous(), res
                    y_small = ops.linear(big[:m].contiguous(), w, **kw_s)   # <= 2 x 256 workgroups: small-grid form
                    y_ring = ops.linear(big, w, **kw_b)[:m]                 # 1250 row tiles: ring kernel
                    assert torch.equal(y_small, y_ring), (K, N, m, sorted(kw))
        # a whole encoder layer (fused q|k|v projection writing the attention kernel's operand slots, attention, tail): one sequence
        # alone against the same sequence as element 0 / 5 of a batch of 6
        layer = GenericEncoderLayer(model_dim=256, head_dim=32, att_mode="self", att_type="full").to(gpu).eval()
        x = torch.randn(6, 4800, 256, generator=g).to(gpu)
        y6 = layer(x)
        assert torch.equal(layer(x[:1].contiguous())[0], y6[0]) and torch.equal(layer(x[5:].contiguous())[0], y6[5])
        cross = GenericEncoderLayer(model_dim=256, context_dim=256, head_dim=32, att_mode="cross", att_type="full").to(gpu).eval()
        c = torch.randn(6, 4800, 256, generator=g).to(gpu)
        z6 = cross(x, c)
        assert torch.equal(cross(x[2:3].contiguous(), c[2:3].contiguous())[0], z6[2])
    finally:
        nerfmatch_amd.set_precision("fp32")


def test_post_norm_encoder_layers_vs_reference(gpu, built_lib):
    """Round 5: `norm_type="post"` (reference forward_post_norm, attention.py:209-221 -- no shipped yaml selects it; it used to raise):
    self and cross attention against the reference's own outputs, on the fp32 matrix cores and on the split-bf16 path."""
    import nerfmatch_amd

    fx = load_golden("matcher_postnorm")
    for prec in ("fp32", "bf16x3"):
        rng = np.random.default_rng(int(fx["weights_seed"]))
        nerfmatch_amd.set_precision(prec)
        try:
            for mode in ("self", "cross"):
                sd = {}
                synth._encoder_layer(sd, rng, "L", 256, cross=False)
                layer = GenericEncoderLayer(model_dim=256, context_dim=256, head_dim=32, norm_type="post", att_mode=mode)
                assert len(layer.norm1) == 1  # one LayerNorm also in cross mode: the reference's state-dict layout
                layer = load_layer(sd, "L", layer).to(gpu)
                x = fx[f"{mode}_x"].to(gpu)
                y = layer(x) if mode == "self" else layer(x, fx["cross_c"].to(gpu))
                assert maxdiff(y, fx[f"{mode}_y"]) < TOL, (prec, mode)
        finally:
            nerfmatch_amd.set_precision("fp32")


def test_speculative_single_pair_path_equals_the_ordinary_one(gpu, built_lib):
    """Round 5: for single-pair batches the fine stage and the match assembly are issued BEFORE the count read-back, on the first `cap`
    slots of the zero-initialised match list (NeRFMatcherMS._speculate; assembly = one kernel, nm_assemble_matches).  Every output must be
    the ordinary path's, bit for bit -- with a capacity above the count, and with one below it (the fall-back re-runs the fine stage)."""
    import nerfmatch_amd

    fx = load_golden("matcher_c2f")
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], fx["pt_feat"].shape[1]
    mk = lambda: dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu), pt3d=fx["pt3d"].to(gpu),
                      pt_feat=fx["pt_feat"].to(gpu), pt_mask=torch.ones(1, N, dtype=torch.bool, device=gpu), pt2d=fx["pt2d"].to(gpu))
    keys = ("mpt2d_c", "mpt2d_f", "mpt3d", "m_bids", "mconf", "expec_f", "pred_mask")
    for prec in ("fp32", "bf16x3"):
        nerfmatch_amd.set_precision(prec)
        try:
            res = {}
            for tag, spec, top in (("plain", False, None), ("spec", True, None), ("spec_small_cap", True, 1)):
                m = make_c2f(fx, gpu)
                m.keep_conf = False
                m.SPECULATE_SINGLE_PAIR = spec
                if tag == "spec_small_cap":
                    m._spec_cap = lambda M_: 1  # fewer slots than matches: the ordinary fine stage must take over
                d = mk()
                m.forward(d, mutual=True, match_thres=0.0)
                res[tag] = d
            K = res["plain"]["pred_num"]
            assert K > 1
            for tag in ("spec", "spec_small_cap"):
                assert res[tag]["pred_num"] == K
                for k in keys:
                    assert torch.equal(res[tag][k], res["plain"][k]), (prec, tag, k)
                for a, b in zip(res[tag]["match_ids"], res["plain"]["match_ids"]):
                    assert torch.equal(a, b)
        finally:
            nerfmatch_amd.set_precision("fp32")


def test_blob_cache_eviction_keeps_fetched_pointers_valid(gpu, built_lib, monkeypatch):
    """Round 5 (found as an order-dependent failure of the full-size tests): an op that takes several packed-weight blobs -- the encoder tail
    takes three -- fetches their pointers one after the other; the cache used to FREE every blob when it overflowed, so the third fetch's
    allocation could land on the first blob and its pack kernel overwrite weights the launch had not consumed.  With a generation limit of ONE
    entry every fetch evicts: the layer must still return what it returns with an ample cache."""
    import nerfmatch_amd

    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 640, 256, generator=g).to(gpu)
    c = torch.randn(2, 512, 256, generator=g).to(gpu)
    nerfmatch_amd.set_precision("bf16x3")
    try:
        for mode in ("self", "cross"):
            layer = GenericEncoderLayer(model_dim=256, context_dim=256, head_dim=32, att_mode=mode, att_type="full").to(gpu).eval()
            args = (x,) if mode == "self" else (x, c)
            ops.invalidate_caches()
            want = layer(*args)
            monkeypatch.setattr(ops, "_LINEAR_LIMIT", 1)
            ops.invalidate_caches()
            for _ in range(3):  # (every call re-packs every weight: allocation patterns vary)
                got = layer(*args)
                assert torch.equal(got, want), mode
            monkeypatch.setattr(ops, "_LINEAR_LIMIT", 256)
    finally:
        nerfmatch_amd.set_precision("fp32")
        ops.invalidate_caches()


@pytest.mark.parametrize("K,count,C0", [(300, 190, 256), (8, 8, 256), (13, 0, 256), (257, 257, 64), (1, 1, 256)])
def test_fine_pt_proj_one_launch_vs_gather_and_linears(gpu, built_lib, K, count, C0):
    """nm_fine_pt_proj (round 5: the point side of the fine stage in one launch, fp32 FMAs) against gather + two nm_linear launches and
    against fp64: valid slots agree to fp32 rounding, slots behind the device count are zeros, ragged last group, count 0."""
    g = torch.Generator().manual_seed(K + C0)
    src = torch.randn(1000, C0, generator=g).to(gpu)
    ids = torch.randint(0, 1000, (K,), generator=g).to(gpu)
    cnt = torch.tensor([count], dtype=torch.int32, device=gpu)
    lin0, lin1 = torch.nn.Linear(C0, 128).to(gpu), torch.nn.Linear(128, 128).to(gpu)
    assert ops.fine_pt_proj_supported(lin0, lin1)
    got = ops.fine_pt_proj(src, ids, cnt, lin0, lin1)
    assert got.shape == (K, 128)
    assert float(got[count:].abs().max()) == 0.0 if count < K else True
    if count:
        x = src[ids[:count]].double()
        want = (x @ lin0.weight.double().T + lin0.bias.double()) @ lin1.weight.double().T + lin1.bias.double()
        scale = want.abs().max().item()
        assert (got[:count].double() - want).abs().max().item() < 2e-6 * scale
        rows = ops.gather_rows(src, ids, cnt)
        chain = ops.linear(ops.linear(rows, lin0.weight, lin0.bias), lin1.weight, lin1.bias)
        assert (got[:count] - chain[:count]).abs().max().item() < 4e-6 * scale
    # unsupported widths are refused by the predicate (the caller then takes the three-launch path)
    assert not ops.fine_pt_proj_supported(torch.nn.Linear(C0, 64), torch.nn.Linear(64, 64))


def test_layernorm_pair_is_two_layernorms_bit_for_bit(gpu, built_lib):
    """nm_layernorm2 (round 5: both pre-norms of a cross-attention layer in one launch): the same bits as two nm_layernorm calls, ragged
    row counts, different affine parameters; unequal widths fall back to two launches."""
    g = torch.Generator().manual_seed(3)
    for r0, r1, dim in ((4800, 4801, 256), (3, 1, 128), (130, 7, 64)):
        x0, x1 = torch.randn(2, r0, dim, generator=g).to(gpu), (torch.randn(r1, dim, generator=g) * 3 + 1).to(gpu)
        l0, l1 = torch.nn.LayerNorm(dim).to(gpu), torch.nn.LayerNorm(dim, eps=1e-6).to(gpu)
        with torch.no_grad():
            l0.weight.copy_(torch.randn(dim, generator=g)); l0.bias.copy_(torch.randn(dim, generator=g))
            l1.weight.copy_(torch.randn(dim, generator=g)); l1.bias.copy_(torch.randn(dim, generator=g))
        y0, y1 = ops.layernorm_pair(x0, l0, x1, l1)
        assert y0.shape == x0.shape and y1.shape == x1.shape
        assert torch.equal(y0, ops.layernorm(x0, l0.weight, l0.bias, l0.eps)) and torch.equal(y1, ops.layernorm(x1, l1.weight, l1.bias, l1.eps))
        assert (y1 - torch.nn.functional.layer_norm(x1, (dim,), l1.weight, l1.bias, l1.eps)).abs().max().item() < 1e-5 * max(1.0, y1.abs().max().item())
    a, b = ops.layernorm_pair(torch.randn(5, 128, device=gpu), torch.nn.LayerNorm(128).to(gpu), torch.randn(5, 256, device=gpu), torch.nn.LayerNorm(256).to(gpu))
    assert a.shape == (5, 128) and b.shape == (5, 256)




@pytest.mark.parametrize("K,count,B", [(300, 190, 1), (64, 64, 3), (5, 0, 1), (1, 1, 2)])
def test_fine_window_layer_one_launch_vs_generic_kernels(gpu, built_lib, K, count, B):
    """nm_fine_window_layer (round 5: window gather + the fine self-attention encoder layer in one launch on the matrix cores) against the
    window gather followed by the generic layer kernels (LayerNorm, GEMMs, small attention), and against torch's own modules in fp64: windows
    that hang over the map's border (zero padding), several maps, a device count below the slot count, a ragged last workgroup."""
    from nerfmatch_amd.modules.attention import SelfAttentionBlock

    g = torch.Generator().manual_seed(K + B)
    Hf, Wf = 24, 32  # fine map of a 48 x 64 image: coarse cells 6 x 8
    ffeat = torch.randn(B, 128, Hf, Wf, generator=g).to(gpu)
    cells = (Hf // 4) * (Wf // 4)
    i_ids = torch.randint(0, cells, (K,), generator=g).to(gpu)
    i_ids[: min(K, 4)] = torch.tensor([0, Wf // 4 - 1, cells - 1, cells - Wf // 4][: min(K, 4)], device=gpu)  # the four corners
    map_ids = torch.randint(0, B, (K,), generator=g).to(gpu)
    cnt = torch.tensor([count], dtype=torch.int32, device=gpu)
    block = SelfAttentionBlock(1, 128, att_type="full", head_dim=16).to(gpu).eval()
    with torch.no_grad():
        for p in block.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else 0.08) + (1.0 if p.dim() == 1 and p.shape[0] == 128 and False else 0.0))
    assert not ops.fine_window_layer_supported(block, 5, 128)  # (the fp32 setting keeps the generic kernels)
    ops.LINEAR_PRECISION = "bf16x3"
    try:
        assert ops.fine_window_layer_supported(block, 5, 128)
        got = ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4)
    finally:
        ops.LINEAR_PRECISION = "fp32"
    assert got.shape == (K, 25, 128)
    if count:
        win = ops.fine_windows_batch(ffeat, map_ids, i_ids, cnt, 5, 4)
        ref = block(win)  # LINEAR_PRECISION fp32: nm_layernorm + nm_linear (fp32 MFMA) + nm_attention
        scale = ref[:count].abs().max().item()
        assert torch.isfinite(got[:count]).all()
        assert (got[:count] - ref[:count]).abs().max().item() < 2e-5 * scale, ((got[:count] - ref[:count]).abs().max().item(), scale)
        # fp64 evaluation of the same layer with torch ops
        l = block.layers[0]
        x = win[:count].double()
        ln = lambda t, m: torch.nn.functional.layer_norm(t, (128,), m.weight.double(), m.bias.double(), m.eps)
        xh = ln(x, l.norm1[0])
        at = l.attention
        q, k_, v = (xh @ w.weight.double().T for w in (at.proj_q, at.proj_k, at.proj_v))
        sp = lambda t: t.reshape(count, 25, 8, 16).transpose(1, 2)
        att = torch.softmax(sp(q) @ sp(k_).transpose(-1, -2) * at.attend.scale(), -1) @ sp(v)
        a_ = xh + att.transpose(1, 2).reshape(count, 25, 128) @ at.proj_out[0].weight.double().T
        ff = l.feedforward
        h1 = torch.nn.functional.gelu(ln(a_, l.norm2) @ ff.layers[0].weight.double().T + ff.layers[0].bias.double())
        y = xh + h1 @ ff.layers[2].weight.double().T + ff.layers[2].bias.double()
        e64 = (got[:count].double() - y).abs().max().item() / y.abs().max().item()
        print(f"fine window layer K={K} count={count} B={B}: |one launch - generic| {(got[:count] - ref[:count]).abs().max().item() / scale:.2e}, |one launch - fp64| {e64:.2e} of the largest entry")
        assert e64 < 1e-5
        # with the point-side features the kernel returns FineMatching's expectation instead (the layer's output never leaves it)
        pf = torch.randn(K, 128, generator=g).to(gpu)
        ops.LINEAR_PRECISION = "bf16x3"
        try:
            ex = ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_f=pf)
        finally:
            ops.LINEAR_PRECISION = "fp32"
        ex_ref = ops.fine_expectation(pf, ref, cnt, 5)
        assert ex.shape == (K, 3) and torch.isfinite(ex[:count]).all()
        assert bool((ex[count:] == 0).all())  # slots behind the count are written as zeros (round 6), not left as they were
        assert (ex[:count] - ex_ref[:count]).abs().max().item() < 2e-4, (ex[:count] - ex_ref[:count]).abs().max().item()
        ex64 = torch.softmax((pf[:count].double()[:, None] * y).sum(-1) / 128 ** 0.5, -1)
        grid = torch.linspace(-1, 1, 5, dtype=torch.float64)
        gx, gy = grid.repeat(5).to(gpu), grid.repeat_interleave(5).to(gpu)
        assert ((ex64 * gx).sum(-1) - ex[:count, 0].double()).abs().max().item() < 2e-4
        assert ((ex64 * gy).sum(-1) - ex[:count, 1].double()).abs().max().item() < 2e-4
        # ... and with the point side computed inside as well (nm_fine_stage): the same bits as feeding it nm_fine_pt_proj's output
        src = torch.randn(500, 256, generator=g).to(gpu)
        pids = torch.randint(0, 500, (K,), generator=g).to(gpu)
        lin0, lin1 = torch.nn.Linear(256, 128).to(gpu), torch.nn.Linear(128, 128).to(gpu)
        ops.LINEAR_PRECISION = "bf16x3"
        try:
            whole = ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_proj=(src, pids, lin0, lin1))
            two = ops.fine_window_layer(ffeat, map_ids, i_ids, cnt, block, 4, pt_f=ops.fine_pt_proj(src, pids, cnt, lin0, lin1))
        finally:
            ops.LINEAR_PRECISION = "fp32"
        assert torch.equal(whole[:count], two[:count])
    # shapes outside the kernel's: refused by the predicate
    assert not ops.fine_window_layer_supported(SelfAttentionBlock(2, 128, att_type="full", head_dim=16), 5, 128)
    assert not ops.fine_window_layer_supported(SelfAttentionBlock(1, 256, att_type="full", head_dim=32), 5, 256)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", synth.MATCHER_VARIANTS)
def test_option_envelope_vs_reference(gpu, built_lib, name, precision):
    """Round 6: the option values the reference's constructors accept beyond the shipped yamls -- pt_ftype pe3d / pt3d
    with the pt_proj layer (c2f_trainer.py:121-139, :263-287), pt_pe_type "id" (:143-147), the PE in front of the self-attention,
    the coarse model's pt_feat_norm (coarse_trainer.py:42-47, :198-200) -- against the reference's own outputs
    (tests/golden/matcher_envelope.npz): point tokens and confidence 1e-4, identical mutual lists, fine-stage outputs."""
    import nerfmatch_amd
    from nerfmatch_amd.matcher import NeRFMatcherCoarse

    fx = load_golden("matcher_envelope")
    cfg, sd = synth.matcher_variant(name, int(fx["weights_seed"]))
    coarse = name == "coarse_norm"
    m = (NeRFMatcherCoarse if coarse else NeRFMatcherMS)(cfg)
    r = m.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all(k.startswith("im_sa.") for k in r.missing_keys), r
    m.backbone = PrecomputedBackbone(fx["cfeat"].to(gpu), 256) if coarse else PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
    m.to(gpu).eval()
    pf = (fx["feat128"] if name == "nerf128_id" else fx["feat256"])
    M, N = fx["cfeat"].shape[2] * fx["cfeat"].shape[3], pf.shape[1]
    nerfmatch_amd.set_precision(precision)
    try:
        tok = m.extract_pt_feat(pf.clone().to(gpu), fx["pt3d"].clone().to(gpu))
        data = dict(image=torch.zeros(1, 3, 8, 8, device=gpu), im_mask=torch.ones(1, M, dtype=torch.bool, device=gpu), pt3d=fx["pt3d"].clone().to(gpu),
                    pt_feat=pf.clone().to(gpu), pt_mask=torch.ones(1, N, dtype=torch.bool, device=gpu), pt2d=fx["pt2d"].to(gpu))
        m.forward(data, mutual=True) if coarse else m.forward(data, mutual=True, match_thres=0.0)
    finally:
        nerfmatch_amd.set_precision("fp32")
    scale = max(1.0, float(fx[f"{name}_pt_tokens"].abs().max()))
    assert maxdiff(tok, fx[f"{name}_pt_tokens"]) < TOL * scale
    b, i, j = data["match_ids"]
    assert torch.equal(i.cpu(), fx[f"{name}_i_ids"]) and torch.equal(j.cpu(), fx[f"{name}_j_ids"]) and len(i) > 5
    assert maxdiff(data["mconf"], fx[f"{name}_mconf"]) < TOL
    assert maxdiff(data["conf_matrix"], fx[f"{name}_conf"]) < TOL
    if coarse:  # feature_normalization centres the batch's tensors in place (the reference's `x -= centroid`)
        assert maxdiff(data["pt3d"], fx[f"{name}_pt3d_after"]) < 1e-5 and maxdiff(data["pt_feat"], fx[f"{name}_pt_feat_after"]) < 1e-5
    else:
        assert maxdiff(data["expec_f"], fx[f"{name}_expec_f"]) < TOL
        assert maxdiff(data["mpt2d_f"], fx[f"{name}_mpt2d_f"]) < 10 * TOL
        assert maxdiff(data["mpt3d"], fx[f"{name}_mpt3d"]) == 0


def test_option_envelope_training_graph(gpu, built_lib):
    """The same options under autograd.training() (what the iNeRF matching term and the trainer run): the point tokens equal the
    inference path's, and a gradient reaches pt3d through the Fourier description (pe3d) and through the "id" encoding (pt3d_id)."""
    from nerfmatch_amd import autograd as ag

    fx = load_golden("matcher_envelope")
    for name in ("pe3d", "pt3d_id", "nerf128_id"):
        cfg, sd = synth.matcher_variant(name, int(fx["weights_seed"]))
        m = NeRFMatcherMS(cfg)
        m.load_state_dict(sd, strict=False)
        m.backbone = PrecomputedBackbone((fx["cfeat"].to(gpu), fx["ffeat"].to(gpu)), [256, 128])
        m.to(gpu).eval()
        pf = (fx["feat128"] if name == "nerf128_id" else fx["feat256"]).to(gpu)
        ref = m.extract_pt_feat(pf.clone(), fx["pt3d"].to(gpu))
        with torch.enable_grad(), ag.training():
            p3 = fx["pt3d"].to(gpu).clone().requires_grad_(True)
            tok = m.extract_pt_feat(pf.clone(), p3)
            tok.square().sum().backward()
        assert maxdiff(tok.detach(), ref.cpu()) < 2e-4 * max(1.0, float(ref.abs().max()))
        if name != "nerf128_id":
            assert p3.grad is not None and torch.isfinite(p3.grad).all() and float(p3.grad.abs().max()) > 0
