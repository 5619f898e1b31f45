import sys, torch
sys.path.insert(0, "/root/repo")
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for bias_shift in (0.0, 10.0):
    ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128), training=False, stop_layer=3)
    sd = synth.nerf_state_dict(seed=0, density_bias=3.0)
    if bias_shift:
        for k in sd:
            if "pts_linears" in k and k.endswith("bias") or "views_linears.0.bias" in k:
                sd[k] = sd[k] + bias_shift
    ren.load_state_dict(sd); ren.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    n = 256
    xi = torch.zeros(n, 96); xi[:, :90] = torch.rand(n, 90, generator=g) * 2 - 1
    xd = torch.zeros(n, 48); xd[:, :27] = torch.rand(n, 27, generator=g) * 2 - 1
    xi, xd = xi.to(dev), xd.to(dev)
    ops.LINEAR_PRECISION = "fp32"
    ref = inerf.FineField(ren.nerf_fine, dev)
    logit, sig, saved = ref.forward(xi, xd)
    h, hv = saved
    ff = inerf.FusedField(ren.nerf_fine, dev)
    out4, gates = ff.forward(xi, xd)
    print("bias shift", bias_shift, "active fraction per layer", [round(float((x > 0).float().mean()), 3) for x in h], round(float((hv > 0).float().mean()), 3))
    for which in ("logit", "sigma"):
        g_logit = torch.zeros(n, 8, device=dev); g_sig = torch.zeros(n, 8, device=dev)
        if which == "logit": g_logit[:, :3] = torch.randn(n, 3, generator=g).to(dev)
        else: g_sig[:, 0] = torch.randn(n, generator=g).to(dev)
        lin = ops.linear
        # reference pieces
        g_hv = lin(g_logit, ref.WrT, gate=hv)
        g_sig_h = lin(g_sig, ref.WaT)
        gg = lin(lin(g_hv, ref.WvfT), ref.WfT, residual=g_sig_h, gate=h[7])
        skip = None
        for l in range(7, 0, -1):
            if l == 5: skip = lin(gg, ref.W5xT)
            gg = lin(gg, ref.WT[l], gate=h[l - 1])
        x0 = lin(gg, ref.WT[0])
        g4 = torch.cat([g_logit[:, :3], g_sig[:, :1]], 1).contiguous()
        g_xi0, g_xi5, g_xd = inerf._new(n, 96, dev=dev), inerf._new(n, 96, dev=dev), inerf._new(n, 48, dev=dev)
        from nerfmatch_amd._lib import check, dptr, lib, stream
        check(lib().nm_nerf_points_bwd_bf16x3(dptr(ff.blob_bwd, torch.uint8), dptr(g4), dptr(gates, torch.uint8), n, dptr(g_xi0), dptr(g_xi5), dptr(g_xd), stream()), "bwd")
        torch.cuda.synchronize()
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        import ctypes as C
        L = lib()
        L.nm_nerf_points_bwd_bf16x3_dbg.restype = C.c_int
        L.nm_nerf_points_bwd_bf16x3_dbg.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
        g_feat = lin(g_hv, ref.WvfT)
        g7pre = lin(g_feat, ref.WfT, residual=g_sig_h)
        g7 = lin(g_feat, ref.WfT, residual=g_sig_h, gate=h[7])
        g6pre = lin(g7, ref.WT[7]); g6 = lin(g7, ref.WT[7], gate=h[6])
        g5pre = lin(g6, ref.WT[6])
        for stage, want in ((1, g_feat), (2, g7pre), (3, g6pre), (4, g5pre)):
            dbg = torch.zeros(n, 256, device=dev)
            check(L.nm_nerf_points_bwd_bf16x3_dbg(dptr(ff.blob_bwd, torch.uint8), dptr(g4), dptr(gates, torch.uint8), n, dptr(g_xi0), dptr(g_xi5), dptr(g_xd), dptr(dbg), stage, stream()), "dbg")
            torch.cuda.synchronize()
            print(f"      stage {stage}: rel err {rel(dbg, want):.2e}  (|want| max {float(want.abs().max()):.2e}, |got| max {float(dbg.abs().max()):.2e})")
        print(f"   grad through {which}: skip part rel err {rel(g_xi5[:, :90], skip[:, :90]):.2e}  layer-0 part rel err {rel(g_xi0[:, :90], x0[:, :90]):.2e}  xd rel err {rel(g_xd[:, :27], lin(g_hv, ref.WvdT)[:, :27]):.2e}")
