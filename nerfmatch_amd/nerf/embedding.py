"""Encoding modules of nerfmatch/nerf/embedding.py with HIP-backed forward passes.

Inside the render / matcher kernels the encodings are produced in registers as MFMA operands (integrated positional encoding and
view-direction PE in csrc/nerf_fwd*.hip, the 3-D Fourier embedding in csrc/matcher_misc.hip); the modules below serve callers
that reach into the renderer directly, as the reference's iNeRF loop does (`renderer.xyz_encoder(mean, var)`,
`renderer.dirs_encoder(viewdirs)`: nerfmatch_evaluator.py:385-393), through the stand-alone kernels of csrc/encode.hip -- same
signatures, same output layout, no eager fallback.  `PositionalEncodingMIP.scales` is an int64 nn.Parameter in the reference
(embedding.py:58-61) and therefore part of its checkpoints."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


def _rows(t):
    t = t.to(torch.float32)
    return t.reshape(-1, t.shape[-1]).contiguous()


class PositionalEncodingMIP(nn.Module):
    def __init__(self, num_freqs, min_deg=0):
        super().__init__()
        self.min_deg, self.max_deg, self.num_freqs = min_deg, num_freqs, num_freqs
        self.scales = nn.Parameter(torch.tensor([2**i for i in range(min_deg, self.max_deg)]), requires_grad=False)
        self.arith = 0  # 1: the split render kernels' exp2 / fp32-sine device functions (tests pin those against the reference's values)

    def get_embedding_dim(self, in_dim):
        return 2 * in_dim * self.num_freqs + in_dim

    def forward(self, x, y=None):
        """(x_ret, y_ret) for the integrated encoding (y given), sin-PE with the raw input appended otherwise (embedding.py:66-84)."""
        lead, D = x.shape[:-1], x.shape[-1]
        x2 = _rows(x)
        n, F = x2.shape[0], self.max_deg - self.min_deg
        new = lambda w: torch.empty(n, w, device=x2.device, dtype=torch.float32)
        if y is None:
            out = new(2 * F * D + D)
            _lib.check(_lib.lib().nm_mip_encode(_lib.dptr(x2), None, C.c_size_t(n), D, self.min_deg, F, 0, _lib.dptr(out), None, _lib.stream()),
                       "nm_mip_encode")
            return out.reshape(*lead, -1)
        y2 = _rows(y)
        if y2.shape != x2.shape:
            raise ValueError(f"x {tuple(x.shape)} and y {tuple(y.shape)} must have the same shape")
        x_ret, y_ret = new(2 * F * D), new(2 * F * D)
        _lib.check(_lib.lib().nm_mip_encode(_lib.dptr(x2), _lib.dptr(y2), C.c_size_t(n), D, self.min_deg, F, int(self.arith), _lib.dptr(x_ret),
                                            _lib.dptr(y_ret), _lib.stream()), "nm_mip_encode")
        return x_ret.reshape(*lead, -1), y_ret.reshape(*lead, -1)


class FourierEmbedding(nn.Module):
    def __init__(self, num_freqs, logscale=True, scale=1.0):
        super().__init__()
        if not logscale or scale != 1.0:
            raise NotImplementedError("only the log-scale, scale=1 Fourier embedding of the shipped configs is built")
        self.num_freqs, self.logscale, self.scale = num_freqs, logscale, scale

    def get_embedding_dim(self, in_dim):
        return 2 * in_dim * self.num_freqs + in_dim

    def forward(self, x, **kwargs):
        """x (..., D) -> (..., D + 2 D num_freqs) = [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...] (embedding.py:35-46)."""
        lead, D = x.shape[:-1], x.shape[-1]
        x2 = _rows(x)
        out = torch.empty(x2.shape[0], D + 2 * D * self.num_freqs, device=x2.device, dtype=torch.float32)
        _lib.check(_lib.lib().nm_fourier_embed(_lib.dptr(x2), C.c_size_t(x2.shape[0]), D, self.num_freqs, _lib.dptr(out), _lib.stream()),
                   "nm_fourier_embed")
        return out.reshape(*lead, -1)

    def __repr__(self):
        return f"FourierEmbedding(num_freqs={self.num_freqs}, logscale={self.logscale}, scale={self.scale})"
