"""What would a two-way split of the key range buy the attention launches of a ONE-query step?  Emulated with the
shipped kernel: B sequences of S keys against 2B sequences of S/2 keys (the same queries twice, each against one half of the keys) -- the
same MFMA work in twice the wavefronts of half the loop length, WITHOUT the merge pass a real split needs (so: an upper bound of the gain).
Cross attention of a one-query step: B = 1; its self attention (image and point tokens in one batch): B = 2.  Batch 16 for reference."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

from nerfmatch_amd import ops

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
ops.ATTENTION_PRECISION = "bf16x3"
L = S = 4800
g = torch.Generator().manual_seed(0)


def timed(B, Lq, Sk, reps=50):
    q = torch.randn(B, Lq, 256, generator=g).to(dev)
    k = torch.randn(B, Sk, 256, generator=g).to(dev)
    v = torch.randn(B, Sk, 256, generator=g).to(dev)
    for _ in range(5):
        ops.attention(q, k, v, 8, 32**-0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attention(q, k, v, 8, 32**-0.5)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in (1, 2, 16):
    whole, halves = timed(B, L, S), timed(2 * B, L, S // 2)
    print(f"B = {B:2d}: {S} keys {whole:7.1f} us   2 x {S // 2} keys {halves:7.1f} us   difference {whole - halves:+6.1f} us per launch (no merge pass counted)")
