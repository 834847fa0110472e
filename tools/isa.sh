#!/bin/bash
# usage: tools/isa.sh <file.hip> [kernel-name-substring ...]  — compile for gfx950, print resource
# usage and a mnemonic histogram per matching kernel.  Outputs land in /tmp/isa/<stem>/.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$(realpath "$1"); shift
STEM=$(basename "$SRC" .hip)
OUT=/tmp/isa/$STEM; mkdir -p "$OUT"
cd "$OUT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$ROOT/include" -I"$ROOT/cream_amd/csrc" \
  -c "$SRC" -o "$OUT/$STEM.o" -save-temps=obj -Rpass-analysis=kernel-resource-usage > "$OUT/log.txt" 2>&1 || { grep -E "error" -A5 "$OUT/log.txt" | head -40; exit 1; }
S="$OUT/$STEM-hip-amdgcn-amd-amdhsa-gfx950.s"
for pat in "$@"; do
  for k in $(grep -oE "^_Z[A-Za-z0-9_]*${pat}[A-Za-z0-9_]*:" "$S" | tr -d ':' | sort -u); do
    echo "== $k"
    grep -A8 "Function Name: $k" "$OUT/log.txt" | grep -E "VGPRs:|AGPRs|SGPRs:|Scratch|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//' | tr '\n' ';'; echo
    awk -v k="$k" '$0 ~ "^"k":" {f=1} f{print} f&&/s_endpgm/{exit}' "$S" > "$OUT/$k.s"
    grep -oE "^\s+[a-z_0-9]+" "$OUT/$k.s" | sort | uniq -c | sort -rn | head -${TOP:-22}
  done
done
