#!/bin/bash
# A/B builds of libmrt_hip.so with extra -D flags:  tools/build_variant.sh <name> [-DFLAG ...]  ->  tools/_bin/libmrt_<name>.so
# (run a tool against it with MRT_LIB_PATH=tools/_bin/libmrt_<name>.so)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/tools/_bin"
cd "$root/messyerraytracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Xarch_host -mfma "$@" -shared \
	kernels.hip api.hip group.hip device_build.hip host/scene_prep.cpp host/bvh_builder.cpp host/two_level_prep.cpp -o "$root/tools/_bin/libmrt_$name.so" -pthread 2> "$root/tools/_bin/build_$name.log" || { tail -20 "$root/tools/_bin/build_$name.log"; exit 1; }
echo "built tools/_bin/libmrt_$name.so"
