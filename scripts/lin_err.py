import sys; sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from nerfmatch_amd import ops
dev = torch.device("cuda:0")
for M, N, K in [(301, 256, 256), (4800, 768, 256), (19200, 256, 256), (1000, 40, 24)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K**-0.5
    ref64 = (x.double() @ w.double().T)
    out = {}
    for prec in ("fp32", "bf16x3"):
        ops.LINEAR_PRECISION = prec
        out[prec] = ops.linear(x.to(dev), w.to(dev)).cpu().double()
    t32 = F.linear(x, w).double()
    print(M, N, K, "max|err| vs fp64: torch-fp32 %.2e  hip-fp32 %.2e  bf16x3 %.2e   rms bf16x3 %.2e" % ((t32 - ref64).abs().max(), (out["fp32"] - ref64).abs().max(), (out["bf16x3"] - ref64).abs().max(), (out["bf16x3"] - ref64).pow(2).mean().sqrt()))
