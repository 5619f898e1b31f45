#!/bin/bash
# builds A/B variants of the fused kernel into gpurun_out-independent location nerfmatch_amd/lib/variants/
set -e
cd "$(dirname "$0")/.."
mkdir -p nerfmatch_amd/lib/variants
python -m nerfmatch_amd.build >/dev/null
FLAGS="$(python -c 'from nerfmatch_amd.build import FLAGS; print(" ".join(FLAGS))')"   # single source of truth: nerfmatch_amd/build.py
SRC=${NM_SRC:-nerf_fwd}   # which csrc file the -D variants apply to (nerf_fwd | nerf_fwd_bf16)
build() { # name, defines
  /opt/rocm/bin/hipcc $FLAGS $2 -c nerfmatch_amd/csrc/$SRC.hip -o nerfmatch_amd/lib/variants/${SRC}_$1.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Scratch|VGPRs Spill" | sed "s/.*remark: */$1: /" | tr '\n' ' '; echo
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 nerfmatch_amd/lib/variants/${SRC}_$1.o $(ls nerfmatch_amd/lib/*.o | grep -v "/$SRC.o" | grep -v safewait) -o nerfmatch_amd/lib/variants/lib_$1.so
}
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  build "$name" "$defs" &
done
wait
