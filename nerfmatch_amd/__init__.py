"""nerfmatch_amd: MI355X (gfx950) implementation of the NeRFMatch hot path behind the reference's Python API.
See DESIGN.md / INTEGRATION.md; the compute lives in nerfmatch_amd/lib/libnerfmatch_amd.so (include/nerfmatch_amd.h)."""


def set_precision(precision):
    """Select the arithmetic of the matcher's contractions (attention, nn.Linear, similarity GEMM) in one call:
    "bf16x3" = bf16 matrix cores with fp32-accurate hi/lo operand splitting, "fp32" = fp32 MFMA (the default of the
    matcher switches).  The NeRF renderer has its own attribute: `NerfRenderer.precision`, default "fp16x3" -- fp16 hi/lo-split
    operands with pack-time power-of-two scaling, range telemetry and a device-side fp32 fall-back when an operand would
    saturate (nerf/models/nerf.py::NeRF.fused); "bf16x3" / "fp32" / "fp16x1" select the other kernels.  With "bf16x3" here the
    iNeRF refinement runs its fine pass on the fused pointwise kernels (inerf.FusedField)."""
    from . import ops

    if precision not in ("fp32", "bf16x3"):
        raise ValueError(f"precision must be 'fp32' or 'bf16x3', got {precision!r}")
    ops.ATTENTION_PRECISION = ops.LINEAR_PRECISION = ops.MATCH_PRECISION = precision
