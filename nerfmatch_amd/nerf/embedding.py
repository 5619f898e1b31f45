"""Encoding modules kept for state-dict / attribute compatibility with nerfmatch/nerf/embedding.py.

The encodings themselves are evaluated inside the HIP kernels (integrated positional encoding and view-direction PE
in csrc/nerf_fwd.hip, the 3-D Fourier embedding in csrc/matcher_misc.hip).  `PositionalEncodingMIP.scales` is an
int64 nn.Parameter in the reference (embedding.py:58-61) and therefore part of its checkpoints."""
import torch
import torch.nn as nn


class PositionalEncodingMIP(nn.Module):
    def __init__(self, num_freqs, min_deg=0):
        super().__init__()
        self.min_deg, self.max_deg, self.num_freqs = min_deg, num_freqs, num_freqs
        self.scales = nn.Parameter(torch.tensor([2**i for i in range(min_deg, self.max_deg)]), requires_grad=False)

    def get_embedding_dim(self, in_dim):
        return 2 * in_dim * self.num_freqs + in_dim


class FourierEmbedding(nn.Module):
    def __init__(self, num_freqs, logscale=True, scale=1.0):
        super().__init__()
        if not logscale or scale != 1.0:
            raise NotImplementedError("only the log-scale, scale=1 Fourier embedding of the shipped configs is built")
        self.num_freqs, self.logscale, self.scale = num_freqs, logscale, scale

    def get_embedding_dim(self, in_dim):
        return 2 * in_dim * self.num_freqs + in_dim
