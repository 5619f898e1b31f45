import sys, torch
sys.path.insert(0, "/root/repo")
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
g = torch.Generator().manual_seed(1)
n = 128 * 600
xi = torch.zeros(n, 96); xi[:, :90] = torch.rand(n, 90, generator=g) * 2 - 1
xd = torch.zeros(n, 48); xd[:, :27] = torch.rand(n, 27, generator=g) * 2 - 1
xi, xd = xi.to(dev), xd.to(dev)
ops.LINEAR_PRECISION = "fp32"
ref = inerf.FineField(ren.nerf_fine, dev)
logit, sig, saved = ref.forward(xi, xd)
ff = inerf.FusedField(ren.nerf_fine, dev)
out4, gates = ff.forward(xi, xd)
g_logit = torch.zeros(n, 8, device=dev); g_logit[:, :3] = torch.randn(n, 3, generator=g).to(dev)
g_sig = torch.zeros(n, 8, device=dev); g_sig[:, 0] = torch.randn(n, generator=g).to(dev)
gxi_ref, gxd_ref = ref.backward(g_logit, g_sig, saved)
g4 = torch.cat([g_logit[:, :3], g_sig[:, :1]], 1).contiguous()
for rep in range(2):
    gxi, gxd = ff.backward(g4, gates)
    gxi = gxi[0] + gxi[1]
    torch.cuda.synchronize()
    e_d = (gxd - gxd_ref).abs().reshape(600, 128, -1).amax((1, 2)) / gxd_ref.abs().max()
    e_i = (gxi - gxi_ref).abs().reshape(600, 128, -1).amax((1, 2)) / gxi_ref.abs().max()
    bad_d, bad_i = (e_d > 1e-3).nonzero().flatten().tolist(), (e_i > 2e-2).nonzero().flatten().tolist()
    print(f"run {rep}: g_xd bad tiles {len(bad_d)}: {bad_d[:20]} ... max {float(e_d.max()):.2e};  g_xi bad tiles {len(bad_i)}: {bad_i[:20]} max {float(e_i.max()):.2e}, median tile err {float(e_i.median()):.2e}")
fe = (out4[:, 3] - sig[:, 0]).abs().reshape(600, 128).amax(1)
print("forward sigma err per tile: max", float(fe.max()), "bad tiles", (fe > 1e-4).nonzero().flatten().tolist()[:10])
# are the "bad" tiles exactly those where a ReLU gate differs between the two forward evaluations (near-zero activations)?
h, hv = saved
G = gates.cpu().view(torch.int32).reshape(-1, 9, 256, 4)
nrow = lambda r, hi: (r & 3) + 8 * (r >> 2) + 4 * hi
import numpy as np
tid = np.arange(256); lane = tid & 63; wave = tid >> 6; s_ = lane & 31; hi_ = lane >> 5; samp = wave * 32 + s_
def decode(tile, l):
    out = np.zeros((128, 256), bool)
    g_ = G[tile, l].numpy().astype(np.int64) & 0xffffffff
    for u in range(16):
        byte = (g_[:, u >> 2] >> (8 * (u & 3))) & 0xff
        ob, m = u >> 1, u & 1
        for e in range(8):
            bit = (4 + (e >> 1)) if (e & 1) else (e >> 1)
            out[samp, 32 * ob + nrow(8 * m + e, hi_)] = ((byte >> bit) & 1).astype(bool)
    return out
def decode_v(tile):
    out = np.zeros((128, 128), bool)
    g_ = G[tile, 8].numpy().astype(np.int64) & 0xffffffff
    for ob in range(4):
        for r in range(16):
            out[samp, 32 * ob + nrow(r, hi_)] = ((g_[:, ob >> 1] >> (16 * (ob & 1) + r)) & 1).astype(bool)
    return out
for tile in sorted(set(bad_d[:6] + bad_i[:6] + [0, 1, 2, 300])):
    flips = [int((decode(tile, l) != (h[l][tile * 128:(tile + 1) * 128] > 0).cpu().numpy()).sum()) for l in range(8)]
    fv = int((decode_v(tile) != (hv[tile * 128:(tile + 1) * 128] > 0).cpu().numpy()).sum())
    print(f"tile {tile}: gate flips vs the GEMM-chain forward per layer {flips} views {fv}   g_xd err {float(e_d[tile]):.1e} g_xi err {float(e_i[tile]):.1e}")
