#!/bin/bash
# Per-kernel code-object metadata (VGPRs, AGPRs, SGPRs, LDS, scratch, spills) of one translation unit of csrc/, compiled
# device-only with the product flags:    tools/kernel_meta.sh smpl.hip [name filter] [extra hipcc flags...]
set -e
SRC=${1:?source file under csrc/}; FILTER=${2:-.}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/hierarchicalprobabilistic3dhuman_amd/csrc
TMP=$(mktemp -d); trap 'rm -rf $TMP' EXIT
EXTRA=$(python3 - "$SRC" <<PY
import sys
sys.path.insert(0, "$ROOT")
from hierarchicalprobabilistic3dhuman_amd import build
print(" ".join(build.FILE_FLAGS.get(sys.argv[1], [])))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -Wno-unused-function $EXTRA "$@" -I $ROOT/include -I $CS \
    --cuda-device-only -c $CS/$SRC -o $TMP/k.bundle
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/k.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/k.co | awk -v f="$FILTER" '
  /\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.sgpr_count:/ {s=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {vs=$2; if (n ~ f) printf "%-100s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spills v%d s%d\n", substr(n,1,100), v, a, s, l, p, vs, ss}'
