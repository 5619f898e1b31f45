"""GPU-side cost of a chain of small dependent launches: stream launches with the host far ahead (the queue is filled behind a spin
kernel) against a captured hipGraph of the same chain.  What a one-query localisation step pays per launch (83 of them)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
ops.LINEAR_PRECISION = "bf16x3"
x = torch.randn(4800, 256, device=dev)
gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
w = torch.randn(256, 256, device=dev) * 0.05
N = 60


def chain_ln(y):
    for _ in range(N):
        y = ops.layernorm(y, gam, bet)
    return y


def chain_gemm(y):
    for _ in range(N):
        y = ops.linear(y, w)
    return y


for name, chain in (("layernorm 4800x256", chain_ln), ("linear 4800x256x256", chain_gemm)):
    chain(x)
    torch.cuda.synchronize()
    res = []
    for rep in range(5):
        torch.cuda._sleep(int(8e6))  # ~4 ms of GPU spin: the host enqueues the whole chain meanwhile
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        chain(x)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / N * 1e3)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = chain(x)
    g.replay()
    torch.cuda.synchronize()
    gr = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        gr.append(e0.elapsed_time(e1) / N * 1e3)
    print(f"{name}: stream (host ahead) {min(res):.2f} us per launch   hipGraph replay {min(gr):.2f} us per launch")
