"""Builds nerfmatch_amd/lib/libnerfmatch_amd.so from csrc/*.hip with hipcc for gfx950 (in-tree, no JIT cache).

    python -m nerfmatch_amd.build [--force] [--verbose] [--safe]

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


SAFE_LIB = LIBDIR / "libnerfmatch_amd_safewait.so"  # checker build: every counted s_waitcnt vmcnt(n) is vmcnt(0) (csrc/common.h)
# sources whose kernels use counted waits: only these differ in the checker build, the rest is linked from the product objects
SAFE_SOURCES = ("attention_v2", "attention_bwd_v2", "attention_fp8", "encoder_tail", "encoder_tail_bwd", "gemm_bf16", "match_fused", "nerf_fwd_bf16")


SAFE_STAMP = LIBDIR / "build_safewait.stamp"


def build(force=False, verbose=False, safe=False):
    """Product library; with safe=True also the checker library (every counted wait a full wait), which only
    tests/test_safe_wait_gpu.py loads -- it has its own stamp, so the product library's validity does not depend on it and a plain
    build() / first import does not pay for its seven extra objects.  __graft_entry__.build() asks for both."""
    LIBDIR.mkdir(exist_ok=True)
    dig = _digest()
    have_lib = LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == dig
    have_safe = SAFE_LIB.exists() and SAFE_STAMP.exists() and SAFE_STAMP.read_text().strip() == dig
    if not force and have_lib and (have_safe or not safe):
        return LIB
    do_lib, do_safe = force or not have_lib, safe and (force or not have_safe)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, safe_objs = [], []
    procs = []
    for src in sources():
        obj = LIBDIR / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        if do_lib or not obj.exists():
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
        if not do_safe:
            continue
        if src.stem in SAFE_SOURCES:
            sobj = LIBDIR / (src.stem + ".safewait.o")
            procs.append((src, subprocess.Popen([hipcc, *FLAGS, "-DNM_SAFE_WAIT", "-c", str(src), "-o", str(sobj)], stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT, text=True)))
            safe_objs.append(sobj)
        else:
            safe_objs.append(obj)
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
    for lib, ob, do, stamp in ((LIB, objs, do_lib, STAMP), (SAFE_LIB, safe_objs, do_safe, SAFE_STAMP)):
        if do:
            subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, ob), "-o", str(lib)], check=True)
            stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    lib = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, safe="--safe" in sys.argv)
    print(lib)
