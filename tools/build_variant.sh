#!/bin/bash
# A library variant for same-box A/B runs: sai_amd/lib/ab/<name>/libsaihip.so = the current tree's objects
# with <unit> (default windows.hip) taken from a git revision or a file.  Run with SAI_AMD_LIB=<that path>.
# Usage: bash tools/build_variant.sh <name> <git-rev | path-to-file> [unit]
set -e
NAME=$1; SRC=$2; UNIT=${3:-windows.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/sai_amd/lib/ab/$NAME
mkdir -p $OUT/src
cp $ROOT/sai_amd/csrc/*.hpp $OUT/src/
if [ -f "$SRC" ]; then cp "$SRC" $OUT/src/$UNIT; else git -C $ROOT show $SRC:sai_amd/csrc/$UNIT > $OUT/src/$UNIT; fi
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()"
STEM=${UNIT%.*}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -I$ROOT/include -I$OUT/src -c $OUT/src/$UNIT -o $OUT/$STEM.o
OBJS=$(ls $ROOT/sai_amd/lib/obj/*.o | grep -v "/$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/$STEM.o -lz -lpthread -ldl -o $OUT/libsaihip.so
echo $OUT/libsaihip.so
