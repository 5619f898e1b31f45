import sys, torch
sys.path.insert(0, "/root/repo")
from nerfmatch_amd import inerf, ops, synth
from nerfmatch_amd.nerf.renderer import NerfRenderer
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=128), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0)); ren.to(dev).eval()
g = torch.Generator().manual_seed(1)
n = 128
xi = torch.zeros(n, 96); xi[:, :90] = torch.rand(n, 90, generator=g) * 2 - 1
xd = torch.zeros(n, 48); xd[:, :27] = torch.rand(n, 27, generator=g) * 2 - 1
xi, xd = xi.to(dev), xd.to(dev)
ops.LINEAR_PRECISION = "fp32"
ref = inerf.FineField(ren.nerf_fine, dev)
logit, sig, (h, hv) = ref.forward(xi, xd)
ff = inerf.FusedField(ren.nerf_fine, dev)
out4, gates = ff.forward(xi, xd)
torch.cuda.synchronize()
G = gates.cpu().view(torch.int32).reshape(-1, 9, 256, 4)[0]  # [layer][tid][4 dwords]
nrow = lambda r, hi: (r & 3) + 8 * (r >> 2) + 4 * hi
for l in range(8):
    want = (h[l] > 0).cpu()
    got = torch.zeros(n, 256, dtype=torch.bool)
    for tid in range(256):
        lane, wave = tid & 63, tid >> 6
        s, hi = lane & 31, lane >> 5
        sample = wave * 32 + s
        for u in range(16):
            byte = (int(G[l, tid, u >> 2]) >> (8 * (u & 3))) & 0xff
            ob, m = u >> 1, u & 1
            for e in range(8):
                bit = (4 + (e >> 1)) if (e & 1) else (e >> 1)
                got[sample, 32 * ob + nrow(8 * m + e, hi)] = bool((byte >> bit) & 1)
    mism = (got != want)
    print(f"layer {l}: gate mismatches {int(mism.sum())} of {mism.numel()}  (active want {float(want.float().mean()):.3f} got {float(got.float().mean()):.3f})", "units with mismatches:", sorted(set(((mism.nonzero()[:, 1] % 32) // 16 + 2 * (mism.nonzero()[:, 1] // 32)).tolist()))[:16])
# which bit permutation? expected 8 gates of (tid, unit) in element order e = 0..7 vs the stored byte
l = 0
want = (h[l] > 0).cpu()
import collections
cnt = collections.Counter()
for tid in range(0, 256, 7):
    lane, wave = tid & 63, tid >> 6
    s, hi = lane & 31, lane >> 5
    sample = wave * 32 + s
    for u in range(16):
        byte = (int(G[l, tid, u >> 2]) >> (8 * (u & 3))) & 0xff
        ob, m = u >> 1, u & 1
        exp = [int(want[sample, 32 * ob + nrow(8 * m + e, hi)]) for e in range(8)]
        got = [(byte >> b) & 1 for b in range(8)]
        if tid < 8 and u < 4:
            print("tid", tid, "unit", u, "expected (element order)", exp, "stored bits 0..7", got)
        for b in range(8):
            for e in range(8):
                if got[b] == exp[e]:
                    cnt[(b, e)] += 1
tot = len(range(0, 256, 7)) * 16
print("agreement matrix rows = stored bit, cols = element (fraction equal):")
for b in range(8):
    print(b, [round(cnt[(b, e)] / tot, 2) for e in range(8)])
