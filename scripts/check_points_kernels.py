import sys, torch
sys.path.insert(0, "/root/repo")
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
for variant in ("7scenes", "cambridge"):
    app = variant == "cambridge"
    ren = NerfRenderer(synth.nerf_config(variant, num_pts=128), num_frames=8 if app else None, training=False, stop_layer=3)
    ren.load_state_dict(synth.nerf_state_dict(seed=0, app_vocab=8 if app else 0, density_bias=3.0)); ren.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    for n in (1000, 128 * 300 + 37):
        xi = torch.zeros(n, 96); xi[:, :90] = torch.rand(n, 90, generator=g) * 2 - 1
        xd = torch.zeros(n, 48); xd[:, :43 if app else 27] = torch.rand(n, 43 if app else 27, generator=g) * 2 - 1
        xi, xd = xi.to(dev), xd.to(dev)
        ops.LINEAR_PRECISION = "fp32"
        ref = inerf.FineField(ren.nerf_fine, dev)
        logit, sig, saved = ref.forward(xi, xd)
        ff = inerf.FusedField(ren.nerf_fine, dev)
        out4, gates = ff.forward(xi, xd)
        torch.cuda.synchronize()
        e_l = (out4[:, :3] - logit[:, :3]).abs().max().item(); e_s = (out4[:, 3] - sig[:, 0]).abs().max().item()
        print(variant, n, "fwd: logit err %.2e (max %.2f)  sigma err %.2e (max %.2f)" % (e_l, logit[:, :3].abs().max().item(), e_s, sig[:, 0].abs().max().item()))
        g_logit = torch.zeros(n, 8, device=dev); g_logit[:, :3] = torch.randn(n, 3, generator=g).to(dev) * 1e-4
        g_sig = torch.zeros(n, 8, device=dev); g_sig[:, 0] = torch.randn(n, generator=g).to(dev) * 1e-5
        gxi_ref, gxd_ref = ref.backward(g_logit, g_sig, saved)
        g4 = torch.cat([g_logit[:, :3], g_sig[:, :1]], 1).contiguous()
        gxi, gxd = ff.backward(g4, gates)
        gxi = gxi[0] + gxi[1]
        torch.cuda.synchronize()
        for name, a, b in (("g_xi", gxi, gxi_ref), ("g_xd", gxd, gxd_ref)):
            print("   bwd %s: max err %.2e of max %.2e (rel %.2e)  finite %s" % (name, (a - b).abs().max().item(), b.abs().max().item(), (a - b).abs().max().item() / b.abs().max().item(), bool(torch.isfinite(a).all())))
# timing at the iNeRF size
n = 4800 * 65
xi = torch.rand(n, 96, device=dev); xd = torch.rand(n, 48, device=dev)
ff = inerf.FusedField(ren.nerf_fine, dev)
for _ in range(2):
    out4, gates = ff.forward(xi, xd); ff.backward(out4 * 1e-4, gates)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for _ in range(5): out4, gates = ff.forward(xi, xd)
e[1].record()
for _ in range(5): ff.backward(out4, gates)
e[2].record(); torch.cuda.synchronize()
print("n = %d: fused forward %.3f ms, fused backward %.3f ms" % (n, e[0].elapsed_time(e[1]) / 5, e[1].elapsed_time(e[2]) / 5))
ops.LINEAR_PRECISION = "bf16x3"
ref = inerf.FineField(ren.nerf_fine, dev)
for _ in range(2): logit, sig, saved = ref.forward(xi, xd); ref.backward(logit * 1e-4, sig * 1e-4, saved)
torch.cuda.synchronize()
e[0].record()
for _ in range(3): logit, sig, saved = ref.forward(xi, xd)
e[1].record()
for _ in range(3): ref.backward(logit, sig, saved)
e[2].record(); torch.cuda.synchronize()
print("GEMM chain (bf16x3): forward %.3f ms, backward %.3f ms" % (e[0].elapsed_time(e[1]) / 3, e[1].elapsed_time(e[2]) / 3))
