"""Time of the bf16x3 attention kernel at the matcher's sizes (B x 8 heads x 4800 x 4800, head dim 32); NM_ATTN_V2=1: second generation.
    python scripts/perf_attention.py [B ...]      NM_ATTN_AHEAD=3|5|7 forces the ring depth (default: by grid size)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from nerfmatch_amd import ops

dev = torch.device("cuda:0")
ops.ATTENTION_PRECISION = "bf16x3"
for B in ([int(a) for a in sys.argv[1:]] or [16, 32]):
    L = S = 4800
    qkv = torch.randn(B * L, 768, device=dev)
    for _ in range(2):
        ops.attention_fused(qkv, (0, 256), (256, 512), (512, 768), B, L, S, 8, 32 ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.attention_fused(qkv, (0, 256), (256, 512), (512, 768), B, L, S, 8, 32 ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 4.0 * B * 8 * L * S * 32 * 3
    print(f"B={B}: {ms:.3f} ms per call (incl. kv pre-split) = {ms / B:.4f} ms per pair-layer, {fl / ms / 1e9:.0f} TFLOP/s issued")
