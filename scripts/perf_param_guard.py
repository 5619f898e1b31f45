"""GPU time of one ParamGuard.check() (nm_params_fingerprint) over the c2f matcher's and the NeRF's parameters, and that a one-word write through
.data at any position -- first word, last word, an odd offset inside a 16-byte group -- raises the flag.

    python scripts/perf_param_guard.py
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from nerfmatch_amd import ops, synth
from nerfmatch_amd.bench_match import build_evaluator
from nerfmatch_amd.nerf.renderer import NerfRenderer

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ev, _ = build_evaluator(dev, 480, 640, queries=1)
ren = NerfRenderer(synth.nerf_config("7scenes", num_pts=64), training=False, stop_layer=3)
ren.load_state_dict(synth.nerf_state_dict(seed=0, density_bias=3.0))
ren.to(dev).eval()
for name, mod in (("c2f matcher", ev.model), ("NeRF (coarse + fine)", ren)):
    g = ops.ParamGuard(mod)
    g.check()
    side = ops.side_stream(dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    with torch.cuda.stream(side):
        e0.record()
    for _ in range(n):
        g.check()
    with torch.cuda.stream(side):
        e1.record()
    torch.cuda.synchronize()
    t = g.tables
    u64 = torch.int64
    with torch.cuda.stream(side):
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record()
        for _ in range(n):
            ops.lib().nm_params_fingerprint(ops.dptr(t["ptrs"], u64), ops.dptr(t["words"], u64), ops.dptr(t["bt"], torch.int32), ops.dptr(t["bo"], u64),
                                            len(g.params), t["nblk"], ops.dptr(t["cur"], u64), ops.dptr(t["ref"], u64), ops.dptr(g.flag, torch.int32), 0, ops.stream())
        k1.record()
    torch.cuda.synchronize()
    print(f"   the launch alone, {n} back to back: {k0.elapsed_time(k1) / n * 1e3:.2f} us each")
    nbytes = sum(p.numel() * p.element_size() for p in g.params)
    print(f"{name}: {len(g.params)} tensors, {nbytes / 1e6:.2f} MB, {g.tables['nblk']} workgroups: {e0.elapsed_time(e1) / n * 1e3:.2f} us per check "
          f"(back to back on the side stream), flag {g.flag.tolist()}")
    big = max(g.params, key=lambda p: p.numel())
    for pos in (0, 1, 2, 3, 5, big.numel() - 1, big.numel() - 2, big.numel() // 2 + 1):
        old = big.data.view(-1)[pos].clone()
        big.data.view(-1)[pos] += 1e-3
        g.check()
        torch.cuda.synchronize()
        up = int(g.flag[1])
        big.data.view(-1)[pos] = old
        g.key = None  # new baseline
        g.check()
        torch.cuda.synchronize()
        assert up == 1 and int(g.flag[1]) == 0, (name, pos, up, g.flag.tolist())
    print("   one-word writes at 8 positions of the largest tensor: all noticed")
