"""Quick timing of the matcher half at BASELINE shapes (M = N = 4800 tokens)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import synth, ops
from nerfmatch_amd.matcher import NeRFMatcherMS, NeRFMatcherCoarse

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


T = 4800
x = torch.randn(1, T, 256, device=dev)
w = torch.randn(256, 256, device=dev) / 16
for prec in ("fp32", "bf16x3"):
    ops.LINEAR_PRECISION = prec
    for Mx in (T, 4 * T):
        xx = torch.randn(Mx, 256, device=dev)
        for Nx in (256, 768):
            ww = torch.randn(Nx, 256, device=dev) / 16
            ms = timeit(lambda: ops.linear(xx, ww), n=20)
            print(f"linear {prec} {Mx}x{Nx}x256: {ms*1e3:.1f} us  {2*Mx*Nx*256/ms/1e9:.2f} TFLOP/s")
ops.LINEAR_PRECISION = "fp32"
q = torch.randn(1, T, 256, device=dev)
for prec in ("fp32", "bf16x3"):
    ops.ATTENTION_PRECISION = prec
    for Bq in (1, 4):
        qq = torch.randn(Bq, T, 256, device=dev)
        ms = timeit(lambda: ops.attention(qq, qq, qq, 8, 32**-0.5))
        print(f"attention {prec} B={Bq} 4800x4800 8 heads x 32: {ms:.3f} ms  {Bq*4*T*T*256/ms/1e9:.1f} TFLOP/s (algorithmic)")
ops.ATTENTION_PRECISION = "fp32"
g, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
ms = timeit(lambda: ops.layernorm(x, g, b))
print(f"layernorm 4800x256: {ms*1e3:.1f} us")
im, pt = synth.separated_features(T, T, 256)
im, pt = im.to(dev), pt.to(dev)
for prec in ("fp32", "bf16x3"):
    ops.MATCH_PRECISION = prec
    for conf in (True, False):
        ms = timeit(lambda: ops.dual_softmax_match(im, pt, 10.0, mutual=True, want_conf=conf), n=10)
        print(f"dual_softmax_match {prec} 4800x4800 want_conf={conf}: {ms:.3f} ms")
ops.MATCH_PRECISION = "fp32"
m = NeRFMatcherMS(synth.matcher_config("c2f"))
m.load_state_dict(synth.matcher_state_dict("c2f"), strict=False)
m.to(dev).eval()
img = torch.randn(1, 3, 480, 640, device=dev)
pt_feat = torch.relu(torch.randn(1, T, 256, device=dev))
pt3d = torch.randn(1, T, 3, device=dev)
from nerfmatch_amd.synth import K_7SCENES
def run():
    data = dict(image=img, im_mask=torch.ones(1, T, dtype=torch.bool, device=dev), pt3d=pt3d, pt_feat=pt_feat,
                pt_mask=torch.ones(1, T, dtype=torch.bool, device=dev), pt2d=torch.zeros(1, T, 2, device=dev))
    m.forward(data, mutual=True)
    return data
d = run()
print("c2f matches:", d["match_ids"][0].shape[0])
ms = timeit(run, n=3, warm=1)
print(f"NeRFMatcherMS.forward (stub backbone) 4800 tokens: {ms:.2f} ms")
