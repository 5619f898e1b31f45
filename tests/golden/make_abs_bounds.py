"""Measured absolute maxima (NM_RECORD_ABS_BOUNDS=<file> python -m pytest tests -m gpu, on an MI355X) -> tests/golden/abs_bounds.json:
bound = 1.5 x measured, rounded up to two significant digits, never below 5e-7 (an error of exactly zero still gets a bound a later run's
last-bit difference can live with).  Usage: python tests/golden/make_abs_bounds.py <measured.json> [<measured2.json> ...]"""
import json
import math
import sys
from pathlib import Path


def round_up(x):
    if x <= 0:
        return 5e-7
    e = math.floor(math.log10(x)) - 1
    return max(5e-7, math.ceil(x / 10 ** e) * 10 ** e)


meas = {}
for f in sys.argv[1:]:
    for k, v in json.loads(Path(f).read_text()).items():
        meas[k] = max(meas.get(k, 0.0), v)
out = {k: float(f"{round_up(1.5 * v):.3g}") for k, v in sorted(meas.items())}
dst = Path(__file__).resolve().parent / "abs_bounds.json"
dst.write_text(json.dumps(out, indent=0, sort_keys=True) + "\n")
print(f"{len(out)} bounds -> {dst}")
