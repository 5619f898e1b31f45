"""Builds nerfmatch_amd/lib/libnerfmatch_amd.so from csrc/*.hip with hipcc for gfx950 (in-tree, no JIT cache).

    python -m nerfmatch_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libnerfmatch_amd.so"
STAMP = LIBDIR / "build.stamp"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wno-unused-result"]


def sources():
    return sorted(CSRC.glob("*.hip"))


def _digest():
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [PKG.parent / "include" / "nerfmatch_amd.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    LIBDIR.mkdir(exist_ok=True)
    dig = _digest()
    if not force and LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = LIBDIR / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if verbose or p.returncode:
            sys.stderr.write(out)
        if p.returncode:
            sys.stderr.write(f"hipcc failed on {src}\n")
            failed = True
    if failed:
        raise RuntimeError("nerfmatch_amd: HIP build failed")
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)]
    subprocess.run(cmd, check=True)
    STAMP.write_text(dig)
    return LIB


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(lib)
