#!/bin/bash
# A/B builds of ONE kernel file: profiles/ab_build.sh <tag> <file.hip> [-DFLAG=..]...
# compiles rrmpg_amd/csrc/<file.hip> with the library's own flags plus the
# given ones and links it with the other objects of the current build into
# exp/librrhip_<tag>.so (exp/ is scratch: git-ignored, travels with gpurun).
# profiles/ab_time.py --lib exp/librrhip_<tag>.so times it.
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p exp
make -C rrmpg_amd/csrc -j8 >/dev/null
obj=exp/${src%.hip}_$tag.o
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall \
    -Wno-unused-function "$@" -c rrmpg_amd/csrc/$src -o $obj
others=$(ls rrmpg_amd/csrc/*.o | grep -v "/${src%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o exp/librrhip_$tag.so $obj $others -ldl
echo "built exp/librrhip_$tag.so ($*)"
