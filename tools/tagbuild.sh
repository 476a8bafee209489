#!/bin/bash
# quick experiment build: recompile ONE csrc file with extra flags and link it against the objects of the main build
#   tools/tagbuild.sh <tag> <file.hip> "<-D flags>"   ->  yolo_deepsort_amd/libydsort_<tag>.so  (load with YDS_BUILD_TAG=<tag>)
cd "$(dirname "$0")/../yolo_deepsort_amd"
tag=$1; f=$2; flags=$3
mkdir -p build_$tag
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../include -Icsrc -Wno-unused-result -ffp-contract=off $flags -c csrc/$f -o build_$tag/$f.o || exit 1
objs=$(ls build/*.o | grep -v "/$f.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs build_$tag/$f.o -o libydsort_$tag.so && echo built libydsort_$tag.so
