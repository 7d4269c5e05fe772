#!/bin/bash
# Timing-only build of ONE translation unit with extra -D flags, linked with the product's other objects into
# tools/abl/libnnfac_<tag>.so (NNF_LIBRARY=<that file> selects it; never shipped, git-ignored).
#   bash tools/abl_build.sh <tag> <source.hip> <object it replaces, e.g. k_hals_fast1.o> <flags...>
set -e
TAG=$1; SRC=$2; REPL=$3; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/nn_fac_amd/csrc
mkdir -p $R/tools/abl
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -mllvm -pragma-unroll-threshold=4000000 "$@" -c $C/$SRC -o /tmp/abl_$TAG.o
OBJS=$(make -s -C $C -pn 2>/dev/null | grep '^OBJS' | head -1 | sed 's/^OBJS *= *//' | tr ' ' '\n' | grep -v "^$" | sed "s#^\$(BUILD)#$C/build#;s#^build/#$C/build/#" | grep -v "/$REPL\$" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/abl/libnnfac_$TAG.so $OBJS /tmp/abl_$TAG.o -ldl 2>&1 | tail -3
echo built tools/abl/libnnfac_$TAG.so
