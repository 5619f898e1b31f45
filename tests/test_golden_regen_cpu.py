"""The committed golden fixtures are what tests/golden/make_golden.py produces from the reference TODAY.

Runs only where /root/reference exists (the build container; the GPU box has no reference and skips): regenerates nerf_r32_s32,
matcher_c2f (+ matcher_lsa, matcher_coarse, written by the same function) and matcher_postnorm into a temporary directory by
importing the reference, and compares every array bit for bit -- values, dtypes, key sets -- with the files under tests/golden/."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
REF = Path("/root/reference")


@pytest.mark.skipif(not REF.exists(), reason="the reference tree is only present in the build container")
def test_make_golden_reproduces_the_committed_fixtures(tmp_path):
    env = dict(os.environ, NM_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, str(GOLDEN / "make_golden.py"), "check"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    made = sorted(p.name for p in tmp_path.glob("*.npz"))
    assert made == ["matcher_c2f.npz", "matcher_coarse.npz", "matcher_lsa.npz", "matcher_postnorm.npz", "nerf_r32_s32.npz"], made
    for name in made:
        new, old = np.load(tmp_path / name), np.load(GOLDEN / name)
        assert set(new.files) == set(old.files), (name, set(new.files) ^ set(old.files))
        for k in new.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k, a.dtype, b.dtype, a.shape, b.shape)
            assert a.tobytes() == b.tobytes(), (name, k)
