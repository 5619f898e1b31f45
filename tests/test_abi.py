"""CPU: the C-ABI library loads and exports every symbol include/nerfmatch_amd.h declares (no compute calls)."""
import re
from pathlib import Path

from nerfmatch_amd import _lib

ROOT = Path(__file__).resolve().parents[1]


def header_symbols():
    txt = (ROOT / "include" / "nerfmatch_amd.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nm_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(built_lib):
    h = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 20
    assert h._nm_missing == [], f"declared in _lib.SIGNATURES but not exported: {h._nm_missing}"
    for s in syms:
        assert s in _lib.SIGNATURES, f"{s} is declared in the header but has no ctypes signature"
        assert getattr(h, s) is not None
    assert set(_lib.SIGNATURES) == set(syms)


def test_host_only_entry_points(built_lib):
    h = _lib.lib()
    assert h.nm_abi_version() == 1
    assert h.nm_error_string(0) == b"ok" and b"supported" in h.nm_error_string(2)
    assert h.nm_raygen_count(480, 640, 8) == 4800 and h.nm_raygen_count(480, 480, 8) == 3600
    assert h.nm_nerf_blob_floats() == 611856 and h.nm_nerf_blob_bytes_bf16x3() == 16384 + (143 + 4) * 16384  # 143 K-step slots (feature_linear folded into the views layer) + 4 zero slots of run-ahead padding
    assert h.nm_match_workspace_bytes(4800, 4800, 256) > 4800 * 4800 * 4


def test_argument_validation_without_gpu(built_lib):
    """Bad arguments are rejected before anything is enqueued, so this is safe without a device."""
    import ctypes as C
    h = _lib.lib()
    null = C.c_void_p(0)
    assert h.nm_sample_coarse(null, null, 4, 32, null, null) == 1
    assert h.nm_nerf_fwd(null, null, null, null, 1, 32, 0, 0, -1.0, 0, null, null, null, null, null, null, null, null, null) == 1
    assert h.nm_linear(null, null, null, null, 4, 4, 8, 0, null, null) == 1
