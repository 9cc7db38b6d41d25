#!/bin/bash
# Tuning build of ONE source: the product objects with futuredet_amd/csrc/<src>.hip recompiled with extra defines
# -> tools/probes/libfd_<src>_<tag>.so.  Usage: tools/probes/build_exp.sh fd_conv2d noW -DFD_CONV_EXP=1 [-DFD_V2_TRACE]
# Use: FD_LIB_PATH=tools/probes/libfd_fd_conv2d_noW.so python tools/bf16_conv_layers.py
set -e
cd "$(dirname "$0")/../.."
src=$1; tag=$2; shift 2
path=futuredet_amd/csrc/$src.hip
args=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--src" ]; then path=$2; shift 2; else args+=("$1"); shift; fi
done
set -- "${args[@]}"
python futuredet_amd/build.py > /dev/null
mkdir -p tools/probes/_obj
extra=""
case $src in fd_decode|fd_sweeps|fd_forecast|fd_voxelize) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra "$@" -c $path -o tools/probes/_obj/${src}_$tag.o
objs=$(ls futuredet_amd/csrc/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_${src}_$tag.so $objs tools/probes/_obj/${src}_$tag.o
echo built tools/probes/libfd_${src}_$tag.so
