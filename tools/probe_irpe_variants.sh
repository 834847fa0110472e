#!/bin/bash
# Same-call A/B of compile-time variants of csrc/irpe_attn.hip on the GPU box: every argument is a set of -D flags ("" = the
# committed kernel); irpe_attn.hip is recompiled and the library relinked from the objects that travelled with the snapshot.
#   bash tools/probe_irpe_variants.sh "" "-DIRPE_SCATTER_GROUP=4" "-DIRPE_LQ_WIDE"
# FILE=rpe_index.hip BENCH="python tools/bench_rpe_index.py" probes another source file with another benchmark.
REPO=$(pwd); SUBSETS=${SUBSETS:-k,q,v,qkv}; FILE=${FILE:-irpe_attn.hip}
MLLVM=""; [ "$FILE" = irpe_attn.hip ] && MLLVM="-mllvm -amdgpu-mfma-vgpr-form=1"
for V in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $MLLVM -I$REPO/include -I$REPO/cream_amd/csrc \
    -DCREAM_BUILD_TAG='"probe"' $V -x hip -c cream_amd/csrc/$FILE -o cream_amd/build/$FILE.o || { echo "compile failed: $V"; continue; }
  hipcc -shared -fPIC --offload-arch=gfx950 -o cream_amd/libcream_amd.so cream_amd/build/*.o -lpthread || { echo "link failed"; continue; }
  echo "== variant [$V]"
  if [ -n "$BENCH" ]; then for r in 1 2; do timeout 200 $BENCH 2>/dev/null | grep -E "${GREP:-.}" | cut -c1-${CUT:-600}; done
  else for r in 1 2; do timeout 200 python tools/bench_irpe_terms.py $SUBSETS 2>/dev/null | tr '\n' ' '; echo; done; fi
done
