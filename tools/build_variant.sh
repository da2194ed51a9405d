#!/bin/bash
# usage: [KSRC=sc_match_p] tools/build_variant.sh <name> "<extra hipcc flags for sc_match_h.hip>"  ->  tools/expbuild/libpr_amd_<name>.so
# (experiment builds of the SC matcher; PR_AMD_LIB=<that file> selects it in the Python loader)
set -e
k=${KSRC:-sc_match_h}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/so_dso_place_recognition_amd/csrc
mkdir -p $root/tools/expbuild
make -C $src -s -j8
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -pthread --offload-arch=gfx950 -Wall -Wno-unused-result $2 -c $src/$k.hip -o $root/tools/expbuild/${k}_$1.o
objs=$(ls $src/*.o | grep -v /$k.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o $root/tools/expbuild/libpr_amd_$1.so $objs $root/tools/expbuild/${k}_$1.o -ldl
echo built tools/expbuild/libpr_amd_$1.so
