#!/bin/bash
# Tuning build of ONE source with extra defines, linked with the product's other objects:
#   tools/probes/build_variant.sh <source without .hip> <tag> <defines...>   ->  tools/probes/libfd_<tag>.so
set -e
cd "$(dirname "$0")/../.."
src=$1; tag=$2; shift 2
python futuredet_amd/build.py > /dev/null
mkdir -p tools/probes/_obj
extra=$(python -c "import sys; sys.path.insert(0, 'futuredet_amd'); import build; print(' '.join(build.EXTRA.get('$src.hip', [])))")  # the product's flags for this source
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra "$@" -c futuredet_amd/csrc/$src.hip -o tools/probes/_obj/${src}_$tag.o
objs=$(ls futuredet_amd/csrc/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libfd_$tag.so $objs tools/probes/_obj/${src}_$tag.o
echo built tools/probes/libfd_$tag.so
